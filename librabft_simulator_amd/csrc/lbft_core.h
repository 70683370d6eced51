// lbft_core.h -- the batched LibraBFTv2 discrete-event simulation step, written once for the HIP
// kernels (lbft_hip.hip).  One GPU lane simulates one independent network ("instance").  Where its state
// lives while lbft_k_run executes:
//   * registers: RNG, clock, counters, the rows of the node the current event is for, a small cache of
//     hot block records;
//   * LDS: the front of the event queue (lane-private columns) and the read-only tables of the delay
//     sampler / pacemaker;
//   * HBM: everything -- instance-major for kernel class 0 and the large-network classes (every instance's words contiguous: a lane's
//     node / snapshot / block record is one run of words, read and written with wide accesses at `record base + immediate`), tiles of 64
//     word-interleaved instances for class 1.
//
// The reference keeps records in HashMaps keyed by 64-bit BCS/SipHash values; those hashes are
// identities only, so this model replaces them with structural ids:
//   * one block pool per instance (block id b >= 1; the QC of block b, its state and its ledger
//     snapshot are all identified by b as well: only the block's author ever forms its QC);
//   * per-node knowledge = three node-bitmasks per block (block known / QC in map / state pending);
//   * timeouts, TCs and votes = author bitmasks (+ the timeout's highest_certified_block_round);
//   * a notification = a refcounted snapshot slot holding a handful of ids and masks;
//   * requests/responses are payload-free under the reference semantics (the reference answers a
//     request on the requester itself, so a response can never insert a record -- quirk Q1, asserted by
//     the oracle); with quirks bit 0 they are snapshots of the same kind as notifications.
//
// The file also compiles with a host C++ compiler: oracle/host_model.cpp builds it for the CPU-only
// differential tests (compact model == full-fidelity oracle on thousands of seeds).  That host build
// is test infrastructure; the product library (liblbft_hip.so) contains only the device build.
//
// Reference citations use the SURVEY.md shorthand (simulator.rs = bft-lib/src/simulator.rs, etc.).
#ifndef LBFT_CORE_H
#define LBFT_CORE_H

#include <stdint.h>
#include <math.h>

#include "lbft_math.h"

namespace lbft {

typedef uint8_t u8;
typedef uint32_t u32;
typedef int32_t i32;
typedef uint64_t u64;
typedef int64_t i64;

// Tuning switches of the large-network classes (each measured on the MI355X):
#ifndef LBFT_AX
#define LBFT_AX 1    // author-set extension words staged in registers with the node's fixed rows
#endif
#ifndef LBFT_BX
#define LBFT_BX 1    // a block's node-set extension words fetched once per record copy (lazily, three words in one round trip)
#endif
#ifndef LBFT_SPEC
#define LBFT_SPEC 1  // calendar queue: the entry behind the popped one is fetched ahead
#endif
// ... and for the two-wavefronts-per-SIMD large-network kernels (SimT<K_LARGE_LEAN> / SimT<K_LARGE_EXCHANGE>, 256 registers), where a staged word that does not fit
// is a spilled register and costs more than the round trip it saves.  What pays is decided by what else is in the register file: with
// the trace / free-slot-mask state out of it and ONE cached block record, the staged author sets and the calendar fetch-ahead fit
// (4 / 24 spilled registers with the response bursts of DESIGN.md section 5) -- 16 384 x 64: 386 -> 375 ms, 8 192 x 100: 2.23 -> 2.04 s, live: 3.46 -> 3.32 s, 5.80 -> 5.35 s; the
// blocks' staged node-set words (BX) stay out (+-0), 3 cached records spill 57 / 123 registers (420 ms).
#ifndef LBFT_LEAN_AX
#define LBFT_LEAN_AX 1
#endif
#ifndef LBFT_LEAN_BX
#define LBFT_LEAN_BX 0
#endif
#ifndef LBFT_LEAN_SPEC
#define LBFT_LEAN_SPEC 1
#endif
#ifndef LBFT_LEAN_Q1
#define LBFT_LEAN_Q1 1   // large networks with the record exchange of quirks bit 0 run on a two-wavefronts-per-SIMD kernel too (SimT<K_LARGE_EXCHANGE>): possible
                         // since a response's epochs are separate steps; 16 384 x 64 live: 3.46 s against 4.11 s on the full-register kernel
#endif
#ifndef LBFT_BLK_CACHE_LEAN2
#define LBFT_BLK_CACHE_LEAN2 1
#endif
#ifndef LBFT_BLK_CACHE_LEAN5
#define LBFT_BLK_CACHE_LEAN5 3  // lbft_k_run2l (SimT<K_LARGE_LEAN>): three records fit since the scalar / record accesses stopped holding a register per field
                                // (round 4: 22 spilled registers; c4 354.6 -> 351.6 ms, c5 1.931 -> 1.897 s; two records 386 ms / 2.12 s; the kernel with
                                // the record exchange, SimT<K_LARGE_EXCHANGE>, loses with two: 3.03 against 2.82 s)
#endif
// Kernel class 0 (the headline small-network path): instance-major rows (tile width 1) instead of 64-instance word-interleaved tiles.
// The lanes of a class-0 wavefront work on different nodes, snapshot slots and blocks of their instances, i.e. on different ROWS: in a
// word-interleaved tile every such word access touches one 128-byte line per distinct row (a 41-word node burst: 4 lines per word, a
// snapshot or block record: up to one line per lane and word).  Instance-major, a lane's node / snapshot / block record is one
// contiguous run of words: wide loads, one or two lines per record.
#ifndef LBFT_C0_IMAJOR
#define LBFT_C0_IMAJOR 1
#endif
#ifndef LBFT_C0_HCREG
#define LBFT_C0_HCREG 1   // (lbft_k_run0q, with LBFT_C0_IMAJOR) the node's hcbr buffers ride in the node burst and live in registers
#endif
#ifndef LBFT_FAST_TRUNC_EXP
#define LBFT_FAST_TRUNC_EXP 1  // the delay sampler decides trunc(exp(y)) from a single-precision estimate when that is safe (SimT::trunc_exp)
#endif
#ifndef LBFT_C0_POPC
#define LBFT_C0_POPC 1   // small batches of class-0 networks run lbft_k_run0s (SimT<K_SMALL_WAVE_POP>): the pop's scan by all 64 lanes of the wavefront
#endif
#ifndef LBFT_C0_HOT_FIRST
#define LBFT_C0_HOT_FIRST 1
#endif
#ifndef LBFT_QUAD_PAIR
#define LBFT_QUAD_PAIR 1
#endif
#ifndef LBFT_C0_QUAD
#define LBFT_C0_QUAD 1   // large class-0 batches of 4-node networks with unit rights and log-normal delays run lbft_k_run0q (SimT<K_HEADLINE>)
#endif
// (round 4 built and measured a "light-event drain" -- requests and cancelled timers finished right after the pop, the lane popping again
// before the heavy part of the step: -18 % wavefront-steps, +-0 time in seven variants; EXPERIMENTS.md.  The code is in the history:
// commit "Light-event drain + snapshot hoist as measured variants".)
#ifndef LBFT_C0_NO_TRACE_STATE
#define LBFT_C0_NO_TRACE_STATE 1  // (class 0 never traces) the round trace's bookkeeping -- last_node, vd_time / vd_stamp -- is not maintained in class 0
#endif
#ifndef LBFT_QUAD_LDS_ROUND_TABLES
#define LBFT_QUAD_LDS_ROUND_TABLES 1  // (with the line above: 65 536 x 4 16.43 -> 16.26 ms) lbft_k_run0q: leader / duration lookups that the LDS tables cover are plain LDS reads (not a load through a selected pointer)
#endif
#ifndef LBFT_COMMIT_CHAIN
#define LBFT_COMMIT_CHAIN 1  // (round 4: 65 536 x 4 15.79 -> 15.57 ms) (n <= 32) a commit that does not extend the last one directly: the chain of blocks to commit (nearly always 2-3) is
                             // collected in ONE pass of dependent fetches and committed from registers, instead of one walk per committed block
#endif
#ifndef LBFT_POPC_MAX_LPW
#define LBFT_POPC_MAX_LPW 8u  // networks per wavefront up to which lbft_k_run0s is used (measured: 1 024 x 4: 6.3 against 7.5 ms, 8 192: 10.9 against 12.1,
                              // 16 384 (8 per wavefront): 14.8 against 15.9; 32 per wavefront: the lane-private scan stops at the queue's length and wins)
#endif
#ifndef LBFT_POP_BATCH
#define LBFT_POP_BATCH 16u // packed LDS queue front (class 0): independent loads in flight per batch of the pop's scan
#endif
// ... and for lbft_k_run0q, whose scan is shared by the two 32-lane halves of the wavefront: one pass covers 2 x the batch.  With batches
// of 16 a lane whose queue holds more than 32 events (7 % of the pops, but some lane of 32 in 88 % of the wavefront-steps) sent the
// whole wavefront through a second pass; 24 covers the 48 slots that 99.999 % of the pops stay below.
// lbft_k_run0q: the LDS queue columns are 32 lanes apart whatever the lanes per wavefront (16 or 32), so the column stride is a compile-time
// shift and the slots of a scan batch are immediates of ONE address (24 ds_read_b64 with offsets) instead of two address operations per slot
#ifndef LBFT_QUAD_STRIDE32
#define LBFT_QUAD_STRIDE32 1
#endif
// Record exchange (quirks bit 0): (a) the response's timeout rounds and first set words ride in its first burst instead of a dependent
// fetch behind the chain walk; (b) a certificate of the peer that IS one of the certificates the requester named in its request is known
// by identity -- no record fetch, no cursor: 57 % of a 64-node network's 24.6 k responses per run decide "nothing to walk" that way (host
// model counters, round 4); (c) the proposed block's "known" word from the LDS window when it is there.
#ifndef LBFT_RESP_FAST
#define LBFT_RESP_FAST 1  // (round 4: c4live 2.78 -> 2.71 s, c5live +-0)
#endif
#ifndef LBFT_POP_BATCH_QUAD
#define LBFT_POP_BATCH_QUAD 24u  // (round 4: 15.60 -> 15.42 ms with 48 slots; 72 slots 15.50; batches of 32: 15.87)
#endif
#ifndef LBFT_PACKED_QL_QUAD
#define LBFT_PACKED_QL_QUAD 48u  // LDS queue slots of lbft_k_run0q (a multiple of its batch); the 0.001 % of pushes beyond them spill to the HBM rows
#endif
#define LBFT_MAX_NODES 128  // node / author sets are 1..4 32-bit words (word 0 in the hot rows, the rest in extension rows)

// Sticky per-instance fault bits (readable after the run; never abort the process).
enum Fault : u32 {
  F_QUEUE_OVERFLOW = 1u << 0,
  F_SNAP_OVERFLOW = 1u << 1,
  F_BLOCK_OVERFLOW = 1u << 2,
  F_LOG_OVERFLOW = 1u << 3,
  F_BALLOT_OVERFLOW = 1u << 4,
  F_DURATION_TABLE = 1u << 5,
  F_COMMIT_UNKNOWN_STATE = 1u << 6,  // reference would panic: "Committed states should be known"
  F_COMMIT_NOT_SUCCESSOR = 1u << 7,  // reference would panic in SimulatedContext::commit
  F_STAMP_OVERFLOW = 1u << 8,
  F_INTERNAL = 1u << 9,
  F_STEP_LIMIT = 1u << 10,
  F_TRACE_OVERFLOW = 1u << 11,
  F_EPOCH_OVERFLOW = 1u << 12,  // quirks bit 0 only: more epochs than epoch_archive capacity / sync list overflow  // a node went past round_trace_capacity (round-switch trace only)
};

// Batch-uniform parameters (kernel argument).
struct Params {
  u32 n;       // nodes per instance
  u32 m;       // instances in this batch (on this GPU)
  u32 stride;  // instances the state array holds (m padded to a multiple of 64)
  u32 tw;      // tile width: instances whose rows are word-interleaved (64 for kernel class 0; the lanes per wavefront otherwise)
  u32 rsh;     // log2 of a row's bytes = log2(4 * tw)
  u32 qcap, scap, bcap, lcap;
  i32 max_clock;
  u32 delay_model;  // 0 LogNormal, 1 uniform integer
  double mu, sigma;
  i64 uni_lo;
  u64 uni_span;
  u64 cpe;  // commands_per_epoch
  i64 tci;  // target_commit_interval
  double lambda;
  const u32* weights;  // voting rights (device table; unit_weights short-cuts it)
  u32 mw;              // mask words = ceil(n / 32)
  u32 qheap;           // event queue is a binary heap in the HBM rows (large networks) instead of the LDS-fronted array
  u32 qpack;           // event queue entries are single packed 64-bit words (kernel class 0; set by compute_layout)
  u32 qcal;            // event queue is a calendar (one FIFO per (time, kind) bucket) -- needs max_clock <= LBFT_CAL_MAX_CLOCK
  u32 off_cal_head, cal_chunks /* (the slot that held off_cal_tail until round 6: heads and tails are interleaved now) */, off_cal_bm, cal_buckets;
  u32 total_votes, quorum;
  u32 equiv;         // extension: every node with index % equiv == 0 is an equivocating leader (0 = none; include/lbft.h)
  u32 quirks;                  // bit 1: EpochId::previous() = id - 1 (fixes reference quirk Q2, base_types.rs:31-37); bit 0 unsupported
  u32 drop_ppm, part_size;     // extension "lossy network" (include/lbft.h): random loss per million, partition cut
  i32 part_start, part_end;    // partition active while part_start <= clock < part_end
  u32 unit_weights;  // every voting right is 1 (the reference's SimulatedContext, simulated_context.rs:209-216)
  u32 rot;           // extension: the voting rights of epoch e are weights[(i + e * rot) % n] (include/lbft.h); the leader table then
                     // holds one table of leader_len rounds per shift 0..n-1
  u32 dur_len, leader_len;
  const i64* dur_tab;    // dur_tab[k] = (i64)(delta * pow(k, gamma)) computed by the host libm (pacemaker.rs:123)
  const u8* leader_tab;  // leader_tab[r] = PacemakerState::leader(r), filled by lbft_fill_leader_table
  const u64* exp_tab;
  const u64* zig_x;
  const u64* zig_f;
  // layout (word offsets of the per-instance rows)
  u32 off_node, node_words;
  u32 off_qhi, off_qlo, off_qmeta;
  u32 off_snap, snap_words, off_snap_ref, off_snap_free;
  u32 off_blk, blk_words;
  u32 off_log;
  u32 off_list;  // n > 16 only: receiver list scratch of process_node_actions
  u32 off_arch, ecap;   // quirks bit 0 (Q1 fixed): frozen record stores of past epochs, snapshot format, [n][ecap]
  u32 off_sync;         // quirks bit 0: scratch list of block ids (bcap words) for building a response's record order
  u32 off_rarch, rarch_words;  // lbft_batch_keep_retired_stores: verbatim copy of a node's rows at each epoch change, [n][ecap][rarch_words] (0: off)
  u32 off_trace, rcap;  // round-switch trace (DataWriter, data_writer.rs): first_time[n][rcap] then max_round[n]; rcap == 0: off
  u32 off_ring, ring;   // cooperative large-network kernels: ring of pre-generated RNG draws (ring entries, a power of two; 2 rows each); 0 = none
  u32 ring_topup;       // draws every network's generator runs ahead per event-loop step (0 = only on demand)
  u32 total_words;
  u32 max_steps;  // events per instance per launch (0 = unlimited)
  u32 lpw;        // lanes of each wavefront that carry an instance (1..64): occupancy vs lane-utilisation knob
  u32 ql;         // event-queue slots per instance that live in LDS (slots >= ql spill to the HBM rows)
  u32 blw;        // large-network kernels: entries of the per-network LDS window of block records behind the register cache (a power of two; 0 = none)
  unsigned long long* prof;  // LBFT_PHASE_TIMERS builds only: cycles per phase of the event loop, summed over wavefronts
};

// A wavefront-uniform value that the compiler would otherwise re-read from the kernel-argument segment at every
// use (s_load + s_waitcnt lgkmcnt(0), which also drains the LDS queue): pin it in a vector register instead.
#if defined(__HIP_DEVICE_COMPILE__)
#define LBFT_PIN_VGPR(x) asm volatile("" : "+v"(x))
#else
#define LBFT_PIN_VGPR(x) do { } while (0)
#endif

// Phase timers (diagnostic builds: -DLBFT_PHASE_TIMERS): s_memtime deltas accumulated per phase of the
// event loop, so that one GPU run tells where a wavefront's time goes.  No-ops in product builds.
#define LBFT_NPHASES 32  // [0..29] phases, [30] wavefront loop iterations, [31] last mark (device scratch) / total cycles (host view)
#if defined(LBFT_PHASE_TIMERS) && defined(__HIPCC__)
// Wavefront-level attribution: the first active lane at the mark charges the time since the wavefront's
// previous mark (kept in LDS, wprof[31]) to phase k -- whichever lanes are diverged away at that point.
#define LBFT_MARK(k) do { if (__builtin_amdgcn_mbcnt_hi(__builtin_amdgcn_read_exec_hi(), __builtin_amdgcn_mbcnt_lo(__builtin_amdgcn_read_exec_lo(), 0)) == 0) { \
    u64 now_ = __builtin_readcyclecounter(); wprof[k] += now_ - wprof[31]; wprof[31] = now_; } } while (0)
#define LBFT_COUNT(k) do { if (__builtin_amdgcn_mbcnt_hi(__builtin_amdgcn_read_exec_hi(), __builtin_amdgcn_mbcnt_lo(__builtin_amdgcn_read_exec_lo(), 0)) == 0) wprof[k] += 1; } while (0)
#define LBFT_DRAIN_VMEM() __builtin_amdgcn_s_waitcnt(0x0F70) /* vmcnt(0): attribute the memory drain to the phase that caused it */
#else
#define LBFT_DRAIN_VMEM() do { } while (0)
#define LBFT_MARK(k) do { } while (0)
#define LBFT_COUNT(k) do { } while (0)
#endif
// -DLBFT_RUN_COUNT (with LBFT_PHASE_TIMERS; diagnostic build liblbft_hip_runcount.so): no timers -- the accumulators [0..5] count what the cooperative
// runs of the large-network kernels retire ON THE DEVICE (whose run segments are 64 / lpw lanes wide, not the host build's 64): events retired in
// request / response / notification runs, and the runs of each kind (one per network and pass).  tools/configs.py prints them as `runs_device`;
// tools/configs.py::RUN_SHARES (the roofline's "updates that write timer words only") is taken from them.
#if defined(LBFT_RUN_COUNT) && defined(LBFT_PHASE_TIMERS) && defined(__HIPCC__)
#undef LBFT_MARK
#undef LBFT_DRAIN_VMEM
#define LBFT_MARK(k) do { } while (0)
#define LBFT_DRAIN_VMEM() do { } while (0)
#define LBFT_RUNCNT(slot, n) do { if (n) { atomicAdd(reinterpret_cast<unsigned long long*>(&wprof[slot]), (unsigned long long)(n)); atomicAdd(reinterpret_cast<unsigned long long*>(&wprof[(slot) + 3]), 1ULL); } } while (0)
#else
#define LBFT_RUNCNT(slot, n) do { } while (0)
#endif
// -DLBFT_COOP_PROF (with LBFT_PHASE_TIMERS): phases 6-10 measure the sub-phases of coop_bulk instead of update_node's
#if defined(LBFT_COOP_PROF)
#define LBFT_UMARK(k) do { } while (0)
#define LBFT_CMARK(k) LBFT_MARK(k)
#else
#define LBFT_UMARK(k) LBFT_MARK(k)
#define LBFT_CMARK(k) do { } while (0)
#endif
#if defined(LBFT_HOST_SIG) && !defined(__HIPCC__)
// (tests/tools/divergence_model.cpp: which of the LBFT_STAT points an event passed, one bit each, reported per step)
extern unsigned long long lbft_host_sig;
void lbft_host_step_done();
#define LBFT_STAT(k) (lbft_host_sig |= 1ULL << (k))
#define LBFT_STATN(k, n) do { } while (0)
#define LBFT_STEP_DONE() lbft_host_step_done()
#elif defined(LBFT_HOST_STATS) && !defined(__HIPCC__)
extern unsigned long long lbft_host_stats[64];
#define LBFT_STAT(k) (lbft_host_stats[k]++)
#define LBFT_STATN(k, n) (lbft_host_stats[k] += (n))
#else
#define LBFT_STAT(k) do { } while (0)
#define LBFT_STATN(k, n) do { } while (0)
#endif
#ifndef LBFT_STEP_DONE
#define LBFT_STEP_DONE() do { } while (0)
#endif
// (tests/tools/bucket_stats.cpp, round 6 "Stage A": every popped event and every scheduled one, for the histograms of events per calendar bucket,
// distinct nodes per bucket and zero-delay hazards; nothing in device builds)
#if defined(LBFT_HOST_BUCKETS) && !defined(__HIPCC__)
void lbft_host_pop(int t, unsigned kind, unsigned node, unsigned sender);
void lbft_host_push(long long t, unsigned kind, unsigned node);
#define LBFT_HOOK_POP(t, k, n, s) lbft_host_pop(t, k, n, s)
#define LBFT_HOOK_PUSH(t, k, n) lbft_host_push(t, k, n)
#else
#define LBFT_HOOK_POP(t, k, n, s) do { } while (0)
#define LBFT_HOOK_PUSH(t, k, n) do { } while (0)
#endif
// (tests/tools/mem_lines.cpp, round 6: every access of the event loop to the instance's rows -- instance-relative byte offset, store or load -- for the
// per-structure count of 128-byte lines an event touches; nothing in device builds)
#if defined(LBFT_HOST_MEMHOOK) && !defined(__HIPCC__)
void lbft_host_mem(unsigned byte_offset, int store);
#define LBFT_HOOK_MEM(o, s) lbft_host_mem((unsigned)(o), s)
#else
#define LBFT_HOOK_MEM(o, s) do { } while (0)
#endif

// HBM layout: instances are grouped in tiles of `tw` (the instances one wavefront advances); a tile is contiguous and holds
// its rows word-interleaved: word w of instance i lives at byte (i / tw) * total_words * 4 tw + w * 4 tw + (i % tw) * 4.
// The lanes of a wavefront read row w as one contiguous segment, consecutive rows are adjacent, and everything a wavefront
// ever touches sits in one contiguous window (TLB- and DRAM-page-friendly).  Kernel class 0 (the headline small-network
// path) uses tw = 64 at compile time (two 32-lane wavefronts of a workgroup share a tile: 256-byte rows).  The other
// classes run few networks per wavefront (8 or 16 lanes of 64-node networks, down to one), and there a 64-wide tile
// would make every word access fetch a 128-byte line of which 16-64 bytes are used and spread a 41-word node burst over 41
// lines.  The lanes of such a wavefront work on different nodes (different rows) anyway, so nothing is gained by interleaving
// instances at all: tw = 1, every instance's words contiguous -- a 41-word node burst is 164 contiguous bytes (two or three
// lines instead of 41), a block record one line, a notification's hcbr words a few.
// Inside the run kernel the tile base is wavefront-uniform (an SGPR pair) and a row access is `tile + u32 byte offset`,
// i.e. the saddr + 32-bit voffset form of global_load/global_store: no 64-bit vector address arithmetic.
#define LBFT_ROW_BYTES 256u  // kernel class 0 (tw = 64)

// Instance-level rows.
enum InstField : u32 {
  I_CLOCK = 0, I_STAMP, I_RNG0, I_RNG1, I_RNG2, I_RNG3, I_RNG4, I_RNG5, I_RNG6, I_RNG7,
  I_QLEN, I_SNAP_FREE, I_NBLOCKS, I_FAULT, I_EV0, I_EV1, I_EV2, I_EV3, I_DRAWS, I_DONE,
  I_MAXQ, I_MAXSNAP, I_SNAP_MASK_LO, I_SNAP_MASK_HI, I_LAST_NODE, I_VD_TIME, I_VD_STAMP, I_CAL_CURSOR, I_CAL_FREE, I_CAL_BUMP,
  I_NFOLD /* duplicate timers folded at scheduling time (never queued) */, I_NUPD /* update_node calls */,
  I_RING_HEAD, I_RING_CNT /* ring of pre-generated draws: draws [head, head + cnt) */,
  I_CONT, I_CONT_META /* a response event whose later epochs are still to be inserted (step_begin): next epoch + 1 (0 = none), the event's queue word */, I_WORDS
};

// Node-level rows (RecordStoreState record_store.rs:93-119, PacemakerState pacemaker.rs:60-77,
// NodeState node.rs:28-45, CommitTracker node.rs:50-59, SimulatedNode simulator.rs:53-59,
// SimulatedContext simulated_context.rs:75-83).
// Memory order = the groups of fields end_node writes back together (each group one contiguous run of words: with instance-major
// rows a group is a few wide stores).
enum NodeField : u32 {
  // (set once per epoch)
  NF_STARTUP = 0, NF_EPOCH, NF_INIT_STATE_BLK,
  NF_PREV_EPOCH_HCC,  // commit-certificate block of the previous epoch's record store (quirks bit 1: Q2 fixed)
  // (timer bookkeeping: nearly every event)
  NF_IGNORE_UNTIL,
  NF_LAST_TIMER_T, NF_TIMER_DUPS,  // duplicate-timer folding (see process_node_actions)
  NF_DUP_STAMP,  // creation stamp of the most recently folded duplicate timer (round-switch trace)
  // (the current round of the record store)
  NF_PROPOSED_BLK, NF_CUR_ROUND, NF_TO_MASK, NF_TO_WEIGHT,
  NF_ELECTION,  // 0 ongoing, 1 won, 2 closed; won block in bits 8..31
  NF_BAL0_BLK, NF_BAL0_WEIGHT, NF_BAL0_AUTHORS, NF_BAL1_BLK, NF_BAL1_WEIGHT, NF_BAL1_AUTHORS,
  // (certificates)
  NF_HQC_ROUND, NF_HQC_BLK, NF_HTC_ROUND, NF_HC_ROUND, NF_HCC_BLK, NF_TC_MASK,
  NF_TC_SEL,  // which of the two hcbr[n] buffers holds the timeout certificate (the other: current timeouts)
  // (pacemaker)
  NF_PM_EPOCH, NF_PM_ROUND, NF_PM_LEADER, NF_PM_START, NF_PM_DUR_LO, NF_PM_DUR_HI,
  // (voting constraints, tracker, ledger)
  NF_LVR, NF_LOCKED, NF_LQAT, NF_TR_EPOCH, NF_TR_HCR, NF_TR_LCT,
  NF_NEXT_CMD, NF_LAST_COMMITTED_BLK, NF_NCOMMITS,
  NF_FIXED_WORDS  // followed by the set extension words (n > 32: words 1.. of the four author sets) and hcbr[2][n]: highest_certified_block_round per timeout author
};
// Node row = [fixed words][4 author sets x (mw - 1) extension words][hcbr[2][n]].  Round 6 moved the extension words in front of the hcbr buffers: the
// event loop stages them with the fixed words (begin_node / ax_load), and behind 2n hcbr words they were 0.5-0.8 KB away -- 1.1-1.3 extra 128-byte lines per
// event of a 64- / 100-node network (tests/tools/mem_lines.cpp); now the burst is ONE run of 41 + 4 (mw - 1) words, and the rows of a large network start
// on line boundaries (compute_layout).  Networks of <= 32 nodes have no extension words: their layout is unchanged.
LBFT_HD u32 node_hcbr_off(u32 mw) { return NF_FIXED_WORDS + 4u * (mw - 1u); }

// Block rows.  The first BC_WORDS rows are the "hot record" that the event loop works on (held in a small
// register-resident cache, see Sim::blk_get): the block's round and links, the rounds of its parent and
// grandparent (denormalised at proposal time, so the 3-chain commit rule record_store.rs:221-235 and the
// voting constraints node.rs:256-276 need one record instead of a pointer chase through three), the
// epoch, the ledger depth and the three per-node knowledge masks.  B_TIME / B_CMD are only read when a
// committed history is exported or hashed, B_VOTERS when record hashes are exported (committed_record_hashes).
enum BlockField : u32 {
  B_ROUND = 0, B_LINK /* prev | author << 16 */, B_PREV_ROUND, B_PP /* grandparent | great-grandparent << 16 (block ids) */, B_PP_ROUND, B_EPOCH,
  B_DEPTH /* commands in the ledger after this block */, B_KNOWN, B_QC, B_PEND, BC_WORDS,
  B_TIME = BC_WORDS, B_CMD,
  B_VOTERS,  // authors 0..31 whose votes the block's QuorumCertificate contains (written once, by the author, when it forms the QC)
  B_WORDS
};
#ifndef LBFT_BLK_CACHE_QUAD
#define LBFT_BLK_CACHE_QUAD 4  // lbft_k_run0q: four records fit since round 4 freed the registers (238 VGPRs, no spill): 15.50 -> 15.38 ms; two: 17.4
#endif
#ifndef LBFT_BLK_CACHE
#define LBFT_BLK_CACHE 3  // register-resident block records per instance (plain FIFO); round 1: 2 -> 29.2 ms, 3 -> 27.9, 4 -> 28.0 (19 spilled registers), 5 -> 29.8
#endif

// Snapshot (notification, data_sync.rs:16-39) rows; followed by tc_hcbr[n], to_hcbr[n].
enum SnapField : u32 { S_EPOCH = 0, S_CERTS /* hcc | hqc << 16 */, S_PROP_VOTE /* proposed | vote << 16 */, S_TC_ROUND, S_TO_ROUND, S_TC_MASK, S_TO_MASK, S_FIXED_WORDS };
// Large networks (round 6): bit 31 of S_TC_ROUND / S_TO_ROUND = "an extension word of this set (authors >= 32) is non-zero".  The receiver of a notification /
// response whose set is of its current round fetched those words in a round trip of their own -- nearly always to find them empty (a healthy network has no
// timeouts in flight); the writer has them in registers.  Rounds stay far below 2^31.
#define LBFT_S_XFLAG 0x80000000u

#define LBFT_CAL_MAX_CLOCK 16383  // calendar queue: (max_clock + 1) * 4 buckets per instance (two rows + one bitmap bit each: 0.5 MiB
                                  // per instance at the cap; the host side falls back to the heap when the batch would not fit the HBM)
#define LBFT_NO_LEADER 0xffu
#define LBFT_NEVER INT64_MAX

// Loop unrolling requests of the device build (the host build of the kernel logic leaves its loops to the host compiler)
#if defined(__HIPCC__)
#define LBFT_UNROLL _Pragma("unroll")
#define LBFT_UNROLL4 _Pragma("unroll 4")
#else
#define LBFT_UNROLL
#define LBFT_UNROLL4
#endif
// Branch hints: without them the compiler lays blocks out in source order, i.e. fault handling, capacity spills and once-per-epoch code
// sit in the middle of the event loop; with them they move behind it and the loop's instruction-cache footprint shrinks.
#if defined(LBFT_NO_HINTS)
#define LBFT_UNLIKELY(x) (x)
#define LBFT_LIKELY(x) (x)
#else
#define LBFT_UNLIKELY(x) __builtin_expect(!!(x), 0)
#define LBFT_LIKELY(x) __builtin_expect(!!(x), 1)
#endif
LBFT_RARE u64 udiv64(u64 a, u64 b) { return a / b; }  // (a software division of ~200 instructions on the device; only epoch changes need it)
LBFT_HD u64 rotl64(u64 x, int b) { return (x << b) | (x >> (64 - b)); }

LBFT_HD u64 mulhi64(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul64hi(a, b);
#else
  return (u64)(((unsigned __int128)a * b) >> 64);
#endif
}

LBFT_HD int clz64(u64 x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __clzll((long long)x);
#else
  return __builtin_clzll(x);
#endif
}
LBFT_HD int clz32(u32 x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __clz((int)x);
#else
  return __builtin_clz(x);
#endif
}
LBFT_HD int ctz32(u32 x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __ffs((int)x) - 1;
#else
  return __builtin_ctz(x);
#endif
}

// Rust `f64 as i64` (saturating; NaN -> 0).
LBFT_HD i64 f64_to_i64_sat(double v) {
  if (__builtin_expect(v >= 0.0 && v < 2147483648.0, 1)) return (i64)(i32)v;  // every sane delay: one hardware conversion (truncates toward zero)
  if (v != v) return 0;
  if (v >= 9223372036854775808.0) return INT64_MAX;
  if (v <= -9223372036854775808.0) return INT64_MIN;
  return (i64)v;
}

// ---- Xoshiro256** (rand_xoshiro 0.6) as a value type held in registers ---------------------------
// RING: the kernels whose lanes cooperate on one network (SimT::COOP) keep a ring of pre-generated draws in the instance's
// HBM rows in front of the live generator state: every draw is taken from the ring while it holds any (next_u64), so a
// group of lanes can read the next 64 draws of one network with one load each, evaluate them in parallel, and the
// sequential generator (22 instructions per draw, inherently serial) runs ahead of the consumers in every network of a
// wavefront at once (ring_fill) instead of inside the one lane whose network happens to broadcast.
template <bool RING>
struct RngT {
  u64 s0, s1, s2, s3;
  u32 draws;
  // ring state (RING only): entry e of the ring = rows (rrow + 2e, rrow + 2e + 1) of this instance's column
  char* rtile;
  u32 rbase;         // byte offset of the ring's first row in this lane's column (boff(off_ring))
  u32 rhead, rcnt;   // draws [rhead, rhead + rcnt) are in the ring (indices taken modulo rmask + 1)
  u32 rmask;         // entries - 1 (a power of two); 0xffffffff = no ring attached
  u32 rrsh;          // log2 of a row's bytes in this batch's layout
  LBFT_HD void seed(u64 seed) {  // seed_from_u64: four SplitMix64 outputs
    u64 x = seed, z;
    x += 0x9e3779b97f4a7c15ULL; z = x; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL; z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL; s0 = z ^ (z >> 31);
    x += 0x9e3779b97f4a7c15ULL; z = x; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL; z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL; s1 = z ^ (z >> 31);
    x += 0x9e3779b97f4a7c15ULL; z = x; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL; z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL; s2 = z ^ (z >> 31);
    x += 0x9e3779b97f4a7c15ULL; z = x; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL; z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL; s3 = z ^ (z >> 31);
    draws = 0;
    if (RING) { rhead = 0; rcnt = 0; }
  }
  LBFT_HD u64 step() {
    u64 r = rotl64(s1 * 5, 7) * 9;
    u64 t = s1 << 17;
    s2 ^= s0; s3 ^= s1; s1 ^= s2; s0 ^= s3;
    s2 ^= t;
    s3 = rotl64(s3, 45);
    return r;
  }
  LBFT_HD u32 ring_off(u32 e) const { return rbase + ((e & rmask) << (rrsh + 1)); }  // two rows per entry
  LBFT_HD u64 ring_at(u32 e) const {
    u32 o = ring_off(e);
    LBFT_HOOK_MEM(o, 0); LBFT_HOOK_MEM(o + (1u << rrsh), 0);
    return (u64)*reinterpret_cast<const u32*>(rtile + (size_t)o) | ((u64)*reinterpret_cast<const u32*>(rtile + (size_t)o + (1u << rrsh)) << 32);
  }
  // generator runs ahead: `g` more draws appended to the ring (the caller bounds g by the free room)
  LBFT_HD void ring_fill(u32 g) {
    for (u32 q = 0; q < g; q++) {
      u64 v = step();
      u32 o = ring_off(rhead + rcnt);
      LBFT_HOOK_MEM(o, 1); LBFT_HOOK_MEM(o + (1u << rrsh), 1);
      *reinterpret_cast<u32*>(rtile + (size_t)o) = (u32)v;
      *reinterpret_cast<u32*>(rtile + (size_t)o + (1u << rrsh)) = (u32)(v >> 32);
      rcnt++;
    }
  }
  LBFT_HD u32 ring_room() const { return RING && rmask != 0xffffffffu ? rmask + 1 - rcnt : 0; }
  LBFT_HD u64 next_u64() {
    draws++;
    if (RING) {
      if (rcnt) { u64 v = ring_at(rhead); rhead++; rcnt--; return v; }
    }
    return step();
  }
  // rand 0.8 UniformInt<u64>::sample_single(0, n)  (configuration.rs:67)
  LBFT_HD u64 gen_range_u64(u64 n) {
    u64 zone = (n << clz64(n)) - 1;
    for (;;) {
      u64 v = next_u64();
      u64 lo = v * n;
      if (lo <= zone) return mulhi64(v, n);
    }
  }
  // rand 0.8 UniformInt<u32>::sample_single(0, n) on next_u32 = next_u64 >> 32  (shuffle, simulator.rs:343,370)
  LBFT_HD u32 gen_range_u32(u32 n) {
    u32 zone = (n << clz32(n)) - 1;
    for (;;) {
      u32 v = (u32)(next_u64() >> 32);
      u64 mm = (u64)v * n;
      if ((u32)mm <= zone) return (u32)(mm >> 32);
    }
  }
};
typedef RngT<false> Rng;

// SipHash-1-3 (keys 0,0) of one little-endian u64 == Rust DefaultHasher over `Round(usize)`
// (pacemaker.rs:101-108).
LBFT_HD u64 siphash13_u64(u64 m) {
  u64 v0 = 0x736f6d6570736575ULL, v1 = 0x646f72616e646f6dULL, v2 = 0x6c7967656e657261ULL, v3 = 0x7465646279746573ULL;
#define LBFT_SIPROUND                                                         \
  v0 += v1; v1 = rotl64(v1, 13); v1 ^= v0; v0 = rotl64(v0, 32);              \
  v2 += v3; v3 = rotl64(v3, 16); v3 ^= v2;                                    \
  v0 += v3; v3 = rotl64(v3, 21); v3 ^= v0;                                    \
  v2 += v1; v1 = rotl64(v1, 17); v1 ^= v2; v2 = rotl64(v2, 32);
  v3 ^= m; LBFT_SIPROUND v0 ^= m;
  u64 b = 8ULL << 56;
  v3 ^= b; LBFT_SIPROUND v0 ^= b;
  v2 ^= 0xff;
  LBFT_SIPROUND LBFT_SIPROUND LBFT_SIPROUND
  return v0 ^ v1 ^ v2 ^ v3;
}

// Streaming SipHash-1-3 over u64 words (ledger State = hash of the committed history,
// simulated_context.rs:51-55).
struct Sip13 {
  u64 v0, v1, v2, v3, nwords;
  LBFT_HD void init() {
    v0 = 0x736f6d6570736575ULL; v1 = 0x646f72616e646f6dULL; v2 = 0x6c7967656e657261ULL; v3 = 0x7465646279746573ULL;
    nwords = 0;
  }
  LBFT_HD void word(u64 m) { v3 ^= m; LBFT_SIPROUND v0 ^= m; nwords++; }
  LBFT_HD u64 finish() {
    u64 b = (nwords * 8) << 56;
    v3 ^= b; LBFT_SIPROUND v0 ^= b;
    v2 ^= 0xff;
    LBFT_SIPROUND LBFT_SIPROUND LBFT_SIPROUND
    return v0 ^ v1 ^ v2 ^ v3;
  }
};

// SipHash-1-3 (keys 0,0) over a byte stream: the reference hashes a record as DefaultHasher over "Name::" followed by its BCS
// bytes (smr_context.rs:84-95, simulated_context.rs:238-242).  Only the export of record hashes uses it.
struct SipBytes {
  u64 v0, v1, v2, v3, buf, total;
  u32 nbuf;
  LBFT_HD void init() {
    v0 = 0x736f6d6570736575ULL; v1 = 0x646f72616e646f6dULL; v2 = 0x6c7967656e657261ULL; v3 = 0x7465646279746573ULL;
    buf = 0; total = 0; nbuf = 0;
  }
  LBFT_HD void byte(u32 c) {
    buf |= (u64)(c & 0xffu) << (8 * nbuf);
    total++;
    if (++nbuf == 8) { u64 m = buf; v3 ^= m; LBFT_SIPROUND v0 ^= m; buf = 0; nbuf = 0; }
  }
  LBFT_HD void u64le(u64 v) { for (u32 i = 0; i < 8; i++) byte((u32)(v >> (8 * i))); }  // BCS u64 / usize
  LBFT_HD void uleb(u64 v) {  // BCS sequence length
    while (v >= 0x80) { byte((u32)v | 0x80u); v >>= 7; }
    byte((u32)v);
  }
  LBFT_HD void option(bool some, u64 v) { byte(some ? 1u : 0u); if (some) u64le(v); }  // BCS Option<u64>
  LBFT_HD u64 finish() {
    u64 b = (total << 56) | buf;
    v3 ^= b; LBFT_SIPROUND v0 ^= b;
    v2 ^= 0xff;
    LBFT_SIPROUND LBFT_SIPROUND LBFT_SIPROUND
    return v0 ^ v1 ^ v2 ^ v3;
  }
};
// context.hash(&EpochId(e)) (node.rs:116-118): the initial hash of an epoch's record store
LBFT_HD u64 record_hash_epoch_id(u64 e) {
  SipBytes h; h.init();
  const char name[] = "EpochId::";
  for (u32 i = 0; i < sizeof(name) - 1; i++) h.byte((u32)name[i]);
  h.u64le(e);
  return h.finish();
}
// Block_ (record.rs:51-63): command (proposer, index), time, previous_quorum_certificate_hash, round, author
LBFT_HD u64 record_hash_block(u64 proposer, u64 index, i64 time, u64 prev_qc_hash, u64 round, u64 author) {
  SipBytes h; h.init();
  const char name[] = "Block_::";
  for (u32 i = 0; i < sizeof(name) - 1; i++) h.byte((u32)name[i]);
  h.u64le(proposer); h.u64le(index); h.u64le((u64)time); h.u64le(prev_qc_hash); h.u64le(round); h.u64le(author);
  return h.finish();
}
// Vote_ (record.rs:65-80): epoch_id, round, certified_block_hash, state, committed_state, author
LBFT_HD u64 record_hash_vote(u64 epoch, u64 round, u64 block_hash, u64 state, bool has_cs, u64 cs, u64 author) {
  SipBytes h; h.init();
  const char name[] = "Vote_::";
  for (u32 i = 0; i < sizeof(name) - 1; i++) h.byte((u32)name[i]);
  h.u64le(epoch); h.u64le(round); h.u64le(block_hash); h.u64le(state); h.option(has_cs, cs); h.u64le(author);
  return h.finish();
}

// Timeout_ (record.rs:101-111): epoch_id, round, highest_certified_block_round, author
LBFT_HD u64 record_hash_timeout(u64 epoch, u64 round, u64 hcbr, u64 author) {
  SipBytes h; h.init();
  const char name[] = "Timeout_::";
  for (u32 i = 0; i < sizeof(name) - 1; i++) h.byte((u32)name[i]);
  h.u64le(epoch); h.u64le(round); h.u64le(hcbr); h.u64le(author);
  return h.finish();
}

// EpochConfiguration::pick_author(SipHash13(round)) (configuration.rs:65-75, pacemaker.rs:100-109)
// `shift`: author a holds weights[(a + shift) % n] (rotating voting rights; 0 in the reference)
LBFT_RARE u32 compute_leader(const u32* weights, u32 n, u32 total_votes, u64 round, u32 shift = 0) {
  Rng r;
  r.seed(siphash13_u64(round));
  u64 target = r.gen_range_u64(total_votes);
  u32 i = shift;
  for (u32 a = 0; a < n; a++) {
    if (weights[i] > target) return a;
    target -= weights[i];
    i = i + 1 == n ? 0 : i + 1;
  }
  return 0;  // unreachable
}

// Byte offset of instance i's tile / word offset of a word of instance i in the state array.
LBFT_HD size_t tile_offset_bytes(const Params& p, u32 i) { return (size_t)(i / p.tw) * p.total_words * ((size_t)4 * p.tw); }
LBFT_HD size_t state_words(const Params& p) { return (size_t)p.total_words * p.stride; }  // stride = m padded to 64
LBFT_HD size_t word_offset(const Params& p, u32 i, u32 w) { return tile_offset_bytes(p, i) / 4 + (size_t)w * p.tw + (i & (p.tw - 1u)); }

// ---- lanes of a wavefront cooperating on ONE network (SimT::coop_bulk) -------------------------------------------------
// The cooperative code is written once over "per-lane values": on the device a PL<T> is a register of the executing lane
// and LBFT_FOR_LANES runs its body once, for that lane; the host build (oracle/host_model.cpp, test infrastructure) holds
// all 64 lanes' values in an array and runs each body for lane 0..63 in turn, so that the lane mapping, the ballots and the
// shuffles of the device code are what the CPU-only differential tests execute.  Rules that keep the two equivalent: lanes
// communicate only through the pl_* operations below or through memory written in an EARLIER body, never inside one body.
#if defined(__HIP_DEVICE_COMPILE__)
template <class T> struct PL {
  T v;
  LBFT_HD T& operator[](u32) { return v; }
  LBFT_HD const T& operator[](u32) const { return v; }
};
LBFT_HD u32 lbft_lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
#define LBFT_FOR_LANES(l) for (u32 l = lbft_lane_id(), lbft_once_ = 1; lbft_once_; lbft_once_ = 0)
LBFT_HD u64 pl_ballot(const PL<u32>& p) { return __ballot(p.v != 0); }
LBFT_HD void pl_shfl(PL<u32>& d, const PL<u32>& src, const PL<u32>& lane) { d.v = (u32)__shfl((int)src.v, (int)(lane.v & 63u), 64); }
LBFT_HD u32 pl_read(const PL<u32>& src, u32 k) { return (u32)__builtin_amdgcn_readlane((int)src.v, (int)k); }  // k wavefront-uniform
LBFT_HD void pl_write(PL<u32>& d, u32 k, u32 val) { d.v = lbft_lane_id() == k ? val : d.v; }  // (compare + select: this compiler has no writelane builtin)
#define LBFT_UNI(x, k) ((u32)__builtin_amdgcn_readlane((int)(x), (int)(k)))  // lane k's value of x, in every lane
#define LBFT_IS_LANE(k) (lbft_lane_id() == (k))
#else
template <class T> struct PL {
  T v[64];
  T& operator[](u32 l) { return v[l]; }
  const T& operator[](u32 l) const { return v[l]; }
};
#define LBFT_FOR_LANES(l) for (u32 l = 0; l < 64; l++)
inline u64 pl_ballot(const PL<u32>& p) { u64 m = 0; for (u32 l = 0; l < 64; l++) if (p.v[l]) m |= 1ULL << l; return m; }
inline void pl_shfl(PL<u32>& d, const PL<u32>& src, const PL<u32>& lane) { PL<u32> t = src; for (u32 l = 0; l < 64; l++) d.v[l] = t.v[lane.v[l] & 63u]; }
inline u32 pl_read(const PL<u32>& src, u32 k) { return src.v[k]; }
inline void pl_write(PL<u32>& d, u32 k, u32 val) { d.v[k] = val; }
#define LBFT_UNI(x, k) ((u32)(x))  // the host model runs one network per simulator object: it is its own lane k
#define LBFT_IS_LANE(k) (true)
#endif

struct Actions {  // NodeUpdateActions (interfaces.rs:12-21); should_send has at most one element
  i64 next;
  i32 send_to;  // -1 = none
  bool broadcast;
  bool query_all;
};

// The step's specialisations by name (round-5 review: "magic class numbers"); the values are the template arguments they always were -- the kernels' names
// (lbft_k_run0 / 0q / 0s / 0u / 1l / 2l / 2q, lbft_k_run<1>, lbft_k_run<2>) and machine code are unchanged.  sim_class() returns K_SMALL / K_MID / K_LARGE for a
// batch; lbft_hip.hip picks the specialisation of that class (sim_quad, sim_lean, sim_lean_q1, sim_lean1, the batch's size).
enum KernelClass : int {
  K_SMALL = 0,           // lbft_k_run0:  n <= 16, honest nodes, lossless network, reference routing; packed queue behind the LDS front
  K_MID = 1,             // lbft_k_run<1>: n <= 32, one mask word; every feature
  K_LARGE = 2,           // lbft_k_run<2>: n <= 128, multi-word node / author sets; every feature
  K_GENERIC = 3,         // everything decided at run time: init / finalize / read-back kernels, the node-level C ABI, the host model
  K_LARGE_LEAN = 5,      // lbft_k_run2l: K_LARGE without record exchange, trace and lossy network (sim_lean) -- the cooperative runs live here ...
  K_MID_LEAN = 6,        // lbft_k_run1l: K_MID without them (sim_lean1)
  K_LARGE_EXCHANGE = 7,  // lbft_k_run2q: ... and here: K_LARGE_LEAN plus the record exchange of quirks bit 0 (sim_lean_q1)
  K_SMALL_WAVE_POP = 8,  // lbft_k_run0s: K_SMALL for small batches, the pop's scan by all 64 lanes
  K_HEADLINE = 9,        // lbft_k_run0q: K_SMALL with the headline network (4 nodes, unit rights, log-normal delays) fixed at compile time (sim_quad)
  K_SMALL_UNIFORM = 12   // lbft_k_run0u: K_SMALL_WAVE_POP for ONE network per wavefront, wavefront-uniform code on the scalar unit
};
// ------------------------------------------------------------------------------------------------
// One simulated network: `tile` is the instance's tile, `lane4` the byte offset of its column in a row.
// ------------------------------------------------------------------------------------------------
// CLS specialises the step for a network-size class so that the headline small-network path carries none of
// the large-network machinery:
//   0  n <= 16, all nodes honest, lossless network, no round-switch trace, reference request routing (Q1), array event queue behind the LDS front, receiver list packed in a register, one mask word
//   1  n <= 32, one mask word; heap / packed list decided at run time
//   2  n <= 128, multi-word node/author sets (extension rows), heap event queue, receiver list in HBM rows
//   3  everything decided at run time (init / read-back kernels)
// sim_class() picks the class a batch runs with.
//   5  class 2 without the record exchange of quirks bit 0, the round-switch trace and the lossy network (sim_lean()): the
//      plain large-network path fits 256 registers (21 spilled) and runs two wavefronts per SIMD with half the lanes each
template <int CLS>
struct SimT {
  static constexpr bool LEAN2 = CLS == K_LARGE_LEAN || CLS == K_LARGE_EXCHANGE;  // 7 = 5 plus the record exchange of quirks bit 0 (24 spilled registers; a kernel of its own: with
                                                       // that code compiled in, the runs without it lose 10 %)
  static constexpr bool BIG = CLS == K_LARGE || LEAN2;       // multi-word node / author sets
  static constexpr bool LEAN = LEAN2 || CLS == K_MID_LEAN;      // 6 = class 1 without those three (22 spilled registers at 256)
  // Large networks: the lanes of a wavefront cooperate on one network's broadcasts (coop_bulk); every class that may meet such a
  // batch's state (the generic class 3 reads back / steps any batch) honours its ring of pre-generated draws.
  // 64-wide tiles addressed at compile time for the small-network classes (many lanes per wavefront); the large-network
  // classes address tiles of P.tw = lanes per wavefront (lbft_core.h "HBM layout")
  // 8 = class 0 for SMALL batches (at most LBFT_POPC_MAX_LPW networks per wavefront, lbft_k_run0s): the same step, but the pop's scan of the
  // LDS queue front is done by all 64 lanes of the wavefront (coop_find, run_popc) -- a kernel of its own so that neither carries the
  // other's scan (with both, lbft_k_run0 grew from 57.8 to 61.4 KB and the 65 536-network batch from 22.4 to 24.0 ms: the 64 KB
  // instruction cache again)
  // ... and lbft_k_run0q has the lanes that carry no network -- the upper half of a 32-network wavefront -- scan the other half of their
  // column's slots (coop_find_cols: lane-private early stop at the queue's length kept, the scan's batches split over 64 / lpw lanes)
  // (device only: the host build of the kernel logic -- oracle/host_model.cpp, one network per object -- keeps the lane-private pop)
#if defined(__HIP_DEVICE_COMPILE__)
  static constexpr bool PAIR = CLS == K_HEADLINE && LBFT_QUAD_PAIR != 0;
  static constexpr bool POPC = CLS == K_SMALL_WAVE_POP || CLS == K_SMALL_UNIFORM || PAIR;
  // 12 = class 8 for ONE network per wavefront (lbft_k_run0u), executed as WAVEFRONT-UNIFORM code: the network's index, its rows' base and
  // its LDS columns are the same in all 64 lanes (derived from the wavefront's index through readfirstlane, no lane term), every lane runs
  // the step with the same values, so the compiler's uniformity analysis places the protocol logic on the scalar unit -- SGPR state, scalar
  // compares and real branches instead of per-lane values under execution masks (a lone lane of 64 pays a v_cmp + s_and_saveexec +
  // s_cbranch_execz + s_or per `if`; 64-bit integer work is one scalar instruction instead of two vector ones).  Loads from the (uniform)
  // row addresses stay vector loads (the rows are written in the same loop: the scalar cache is not coherent with them); stores and LDS
  // writes are issued by all lanes with the same address and value.  Only the pop's scan (coop_find) has per-lane values.
  static constexpr bool WUNI = CLS == K_SMALL_UNIFORM;
#else
  static constexpr bool PAIR = false;
  static constexpr bool POPC = false;
  static constexpr bool WUNI = false;
#endif
  // 9 = class 0 with the headline network fixed at compile time (lbft_k_run0q): 4 nodes, unit voting rights, log-normal delays, <= 64
  // snapshot slots, no layout padding -- loop bounds, the quorum, record sizes and the first row offsets become immediates (sim_quad())
  static constexpr bool QUAD = CLS == K_HEADLINE;  // (the small-batch kernel gains nothing from it: 1 024 x 4 6.2 against 5.8 ms, 8 192 x 4 10.0 against 9.9 -- latency-bound)
  static constexpr bool C0 = CLS == K_SMALL || CLS == K_SMALL_WAVE_POP || CLS == K_HEADLINE || CLS == K_SMALL_UNIFORM;
  // (WUNI) the value an out-of-line helper returned, declared wavefront-uniform: a call's result counts as divergent, and through the
  // branches that test it so would every value of the event loop after it (all lanes passed the same arguments)
  LBFT_HD static u32 wuni(u32 v) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (WUNI) return (u32)__builtin_amdgcn_readfirstlane((int)v);
#endif
    return v;
  }
  LBFT_HD static u64 wuni(u64 v) { return WUNI ? ((u64)wuni((u32)(v >> 32)) << 32) | wuni((u32)v) : v; }
  LBFT_HD static double wuni(double v) { return WUNI ? lbft_asdouble(wuni(lbft_asuint64(v))) : v; }
  LBFT_HD u32 NN() const { return QUAD ? 4u : P.n; }
  LBFT_HD u32 MW() const { return QUAD ? 1u : P.mw; }
  LBFT_HD u32 NWORDS() const { return QUAD ? NF_FIXED_WORDS + 8u : P.node_words; }
  // first hcbr word of a node row (behind the set extension words of a large network; a compile-time constant in the small-network classes)
  LBFT_HD u32 HCO() const { return BIG ? NF_FIXED_WORDS + 4u * (MW() - 1u) : CLS == K_GENERIC ? NF_FIXED_WORDS + 4u * (P.mw - 1u) : (u32)NF_FIXED_WORDS; }
  LBFT_HD u32 SWORDS() const { return QUAD ? S_FIXED_WORDS + 8u : P.snap_words; }
  LBFT_HD u32 BWORDS() const { return QUAD ? (u32)B_WORDS : P.blk_words; }
  LBFT_HD u32 OFFNODE() const { return QUAD ? (u32)I_WORDS : P.off_node; }
  LBFT_HD u32 OFFSNAP() const { return P.off_snap; }
  LBFT_HD u32 OFFSREF() const { return P.off_snap_ref; }
  LBFT_HD u32 OFFSFREE() const { return P.off_snap_free; }
  LBFT_HD u32 OFFBLK() const { return P.off_blk; }
  LBFT_HD u32 UNITW() const { return QUAD ? 1u : P.unit_weights; }
  LBFT_HD u32 DMODEL() const { return QUAD ? 0u : P.delay_model; }
  LBFT_HD u32 QUORUM() const { return QUAD ? 3u : P.quorum; }
  LBFT_HD u32 ROT() const { return QUAD ? 0u : P.rot; }
  static constexpr bool C0I = C0 && LBFT_C0_IMAJOR != 0;
  static constexpr bool TILE64 = (C0 && !C0I) || CLS == K_MID || CLS == K_MID_LEAN;
  static constexpr bool HCREG = CLS == K_HEADLINE && C0I && LBFT_C0_HCREG != 0;  // (lbft_k_run0q: 18.7 -> 18.1 ms; no gain in the generic class-0 kernels)
  static constexpr bool IMAJOR = BIG || C0I;  // tile width 1 = every instance's words contiguous (P.tw == 1), addressed at compile time
  static constexpr bool F_AX = LEAN2 ? (LBFT_LEAN_AX != 0) : (LBFT_AX != 0);      // (tuning switches above)
  static constexpr bool F_BX = LEAN2 ? (LBFT_LEAN_BX != 0) : (LBFT_BX != 0);
  static constexpr bool F_SPEC = LEAN2 ? (LBFT_LEAN_SPEC != 0) : (LBFT_SPEC != 0);
  static constexpr bool COOP = BIG;
#ifndef LBFT_HCBR_BATCH
#define LBFT_HCBR_BATCH 8  // hcbr words a lane of a large-network kernel has in flight per round trip when it copies a timeout set (copy_hcbr_to)
#endif
#ifndef LBFT_REQRUN
#define LBFT_REQRUN 1  // (round 6) lbft_k_run2l / lbft_k_run2q: the requests at the head of a bucket are taken by the whole wavefront, a chunk at a time (coop_requests)
#endif
  static constexpr bool REQRUN = (CLS == K_LARGE_LEAN || CLS == K_LARGE_EXCHANGE) && LBFT_REQRUN != 0;
#ifndef LBFT_RUN_MIN
#define LBFT_RUN_MIN 2u  // events at the head of a bucket's chunk from which the whole wavefront takes them as a run
#endif
#ifndef LBFT_RSPRUN
#define LBFT_RSPRUN 1  // (round 6) lbft_k_run2l: runs of responses whose update_node is a no-op are taken by the whole wavefront too (coop_responses)
#endif
  static constexpr bool RSPRUN = REQRUN && CLS == K_LARGE_LEAN && LBFT_RSPRUN != 0;
#ifndef LBFT_RSPRUNQ
#define LBFT_RSPRUNQ 1  // (round 6) lbft_k_run2q: the record exchange's responses that carry nothing their node lacks (response_is_inert) and whose update_node is a no-op, as runs
#endif
  static constexpr bool RSPRUNQ = REQRUN && CLS == K_LARGE_EXCHANGE && LBFT_RSPRUNQ != 0;
  LBFT_HD bool rsp_runs() const { return (RSPRUN && !q1()) || (RSPRUNQ && q1()); }
#ifndef LBFT_NTF_MIN
#define LBFT_NTF_MIN LBFT_RUN_MIN  // events at the head of a bucket's chunk from which a notification run is attempted
#endif
#ifndef LBFT_NTFRUN
#define LBFT_NTFRUN 1  // (round 6) lbft_k_run2l / lbft_k_run2q: runs of notifications that provably leave their node as it was (coop_notifications)
#endif
  static constexpr bool NTFRUN = REQRUN && LBFT_NTFRUN != 0;
#ifndef LBFT_NTFACT
#define LBFT_NTFACT 1  // (round 6) ... including the notifications that only add to their node's current timeouts / ballot (notification_effects)
#endif
  static constexpr bool NTFACT = NTFRUN && F_AX && LBFT_NTFACT != 0;
  static constexpr bool RING = BIG || CLS == K_GENERIC;
  bool coop_on;  // set by run_coop: the event loop is being run by a whole wavefront
  static constexpr u32 PB = CLS == K_HEADLINE ? LBFT_POP_BATCH_QUAD : LBFT_POP_BATCH;  // slots per batch of the packed queue's scan
  LBFT_HD bool coop() const { return COOP && coop_on && P.qcal != 0 && P.ring != 0 && !lossy(); }
  LBFT_HD bool wide() const { return BIG ? true : (CLS == K_GENERIC ? NN() > 32 : false); }
  LBFT_HD bool heap() const { return C0 ? false : (BIG ? true : P.qheap != 0); }
  LBFT_HD bool tracing() const { return !C0 && !LEAN && P.rcap != 0; }  // round-switch trace (DataWriter); class 0 never traces
  LBFT_HD bool q1() const { return !C0 && (!LEAN || CLS == K_LARGE_EXCHANGE) && (P.quirks & 1u) != 0; }  // requests are answered by the peer with real payloads
  LBFT_HD bool cal() const { return !C0 && P.qcal != 0; }
  LBFT_HD bool packed() const { return C0 ? true : (BIG ? false : NN() <= 16); }
  LBFT_HD bool qpacked() const { return C0 ? true : (CLS == K_GENERIC ? P.qpack != 0 : false); }  // one-word queue entries
  const Params& P;
  char* tile;
  u32 lane4;
  // instance scalars cached in registers for the duration of a launch
  i32 clock;
  u32 stamp, qlen, snap_free, nblocks, fault, maxq, maxsnap;
  u32 ev_stamp;   // creation stamp of the event being processed
  u32 cal_cursor, cal_free, cal_bump;  // calendar queue: first possibly non-empty bucket, height of the stack of freed slots, bump allocator
  // calendar queue: the bucket being drained -- its index, its head word (chunk + 1) << 6 | position of the next entry (0 = no bucket open; written back
  // when the bucket is left), the tail word as last seen -- and the next entry, fetched ahead (sp_s1 != 0: sp_meta is valid)
  u32 sp_idx, sp_s1, sp_meta, sp_nx;
  u32 cur_h;
  u32 last_node;  // node of the previous event (round-switch trace)
  // round-switch trace: folded duplicate timers of time vd_time still "pop" in the reference until stamp vd_stamp
  u32 vd_time, vd_stamp;
  u64 snap_mask;  // scap <= 64: free snapshot slots as a bit set held in registers (no free-stack round trip)
  u32 ev0, ev1, ev2, ev3;
  u32 cont;  // quirks bit 0: epoch + 1 at which the response event in I_CONT_META goes on at the next step (0: none)
  u32 n_fold, n_upd;  // duplicate timers folded instead of queued / update_node calls: what the device executes, as opposed to the
                      // reference-equivalent event counts ev0..ev3 (bench.py reports the roofline on both)
  RngT<RING> rng;

  // Front of the event queue: slots [0, ql) live in LDS on the device (lane-private column: element k of
  // this instance is qk[qx(k)], so any per-lane slot index is bank-conflict free); slots >= ql spill to
  // the HBM rows.  The host build (oracle/host_model.cpp) passes plain arrays.  ql == 0: HBM rows only.
  u64* qk;
  u32* qm;
  u32 qstr, qsh, ql;  // column stride (lanes per wavefront, a power of two) and its log2: element k of a column is [k << qsh]
  u32 hsh;            // the same for the hcbr column (attach_hcbr)
#if defined(__HIP_DEVICE_COMPILE__)
  static constexpr bool QS32 = CLS == K_HEADLINE && LBFT_QUAD_STRIDE32 != 0;
#else
  static constexpr bool QS32 = false;  // (the host model passes one plain column per network)
#endif
  LBFT_HD u32 QSH() const { return QS32 ? 5u : qsh; }
  LBFT_HD u32 QSTR() const { return QS32 ? 32u : qstr; }
  LBFT_HD u32 qx(u32 k) const { return k << QSH(); }
  // read-only tables (LDS copies on the device)
  const u64 *zig_x, *zig_f, *exp_tab;
  const u8* leader_lds;   // first leader_lds_len rounds of the leader table
  const i64* dur_lds;     // first dur_lds_len entries of the duration table
  u32 leader_lds_len, dur_lds_len;
#if defined(LBFT_PHASE_TIMERS) && defined(__HIPCC__)
  u64* wprof;  // this wavefront's LDS accumulators
#endif

  LBFT_HD SimT(const Params& p, u32* state, u32 i) : SimT(p, reinterpret_cast<char*>(state) + tile_offset_bytes(p, i), (i & (p.tw - 1u)) * 4u, 0) {}
  LBFT_HD SimT(const Params& p, char* tile_base, u32 lane_byte_offset, int) : P(p), tile(tile_base), lane4(lane_byte_offset), qk(nullptr), qm(nullptr), qstr(0), qsh(0), ql(0), zig_x(p.zig_x), zig_f(p.zig_f), exp_tab(p.exp_tab),
        leader_lds(nullptr), dur_lds(nullptr), leader_lds_len(0), dur_lds_len(0), hc(nullptr), bl(nullptr), bl_n(0), bl_sh(0), plist_lds(nullptr) {
    coop_on = false; cur_xk = 0; wtab = p.weights; hcdirty = 0; sw_epoch = 0; sw_blk = 0;
    if (RING) { rng.rtile = tile; rng.rrsh = rsh(); rng.rbase = boff(P.off_ring); rng.rmask = P.ring ? P.ring - 1u : 0xffffffffu; rng.rhead = 0; rng.rcnt = 0; }
  }
  LBFT_HD void attach_queue(u64* keys, u32* metas, u32 stride, u32 slots) {
    qk = keys; qm = metas; qstr = stride; ql = qpacked() ? slots - slots % PB : slots;  // packed entries are scanned in batches of PB
    qsh = 0;
    while ((1u << qsh) < stride) qsh++;  // (a shift instead of a quarter-rate 32-bit multiply per slot access)
    hsh = qsh;
    // (not in the wavefront-uniform kernel: an inline-asm result counts as divergent, and everything compared with it after it)
    if (!WUNI) LBFT_PIN_VGPR(qsh);
    if (!WUNI) LBFT_PIN_VGPR(ql);
  }
  // highest_certified_block_round buffers of the nodes' timeouts (hcbr[node][2][n], behind the fixed node rows): for
  // networks of <= 4 nodes in kernel class 0 the device keeps them in LDS for the duration of a launch (32 words per
  // instance, lane-private column like the queue), because creating a notification copies them out on nearly every
  // event and a row fetch there is a full memory round trip in the middle of the send loop.
  u32* hc;  // nullptr = the HBM rows
  LBFT_HD void attach_hcbr(u32* column) { hc = column; }
  LBFT_HD bool hc_lds() const { return small_sets() && hc != nullptr && !hc_reg(); }
  // ... or, with instance-major rows, in registers: the 2n words of the event's node lie right behind its fixed words and come with the
  // node burst (every hcbr access is for the node of the current event); indices stay compile-time (value selects).
  mutable u32 hcw[8];
  mutable u32 hcdirty;
  LBFT_HD bool hc_reg() const { return HCREG && NN() <= 4; }
  LBFT_HD void hc_load(u32 nb) const {
    if (!hc_reg()) return;
    hcdirty = 0;
    if (NN() == 4) {
      LBFT_UNROLL
      for (u32 k = 0; k < 8; k++) hcw[k] = ldf(nb, NF_FIXED_WORDS + k);
    } else {
      LBFT_UNROLL
      for (u32 k = 0; k < 8; k++) hcw[k] = (k & 3u) < NN() ? ldf(nb, NF_FIXED_WORDS + (k >> 2) * NN() + (k & 3u)) : 0u;
    }
  }
  LBFT_HD void hc_store(u32 nb) const {
    if (!hc_reg() || !hcdirty) return;
    if (NN() == 4) {
      LBFT_UNROLL
      for (u32 k = 0; k < 8; k++) stf(nb, NF_FIXED_WORDS + k, hcw[k]);
    } else {
      LBFT_UNROLL
      for (u32 k = 0; k < 8; k++) if ((k & 3u) < NN()) stf(nb, NF_FIXED_WORDS + (k >> 2) * NN() + (k & 3u), hcw[k]);
    }
  }
  LBFT_HD u32 hc_get(u32 node, u32 buf, u32 a) const {
    if (hc_reg()) {
      u32 idx = buf * 4u + a;
      u32 v01 = (idx & 1u) ? hcw[1] : hcw[0], v23 = (idx & 1u) ? hcw[3] : hcw[2], v45 = (idx & 1u) ? hcw[5] : hcw[4], v67 = (idx & 1u) ? hcw[7] : hcw[6];
      u32 v03 = (idx & 2u) ? v23 : v01, v47 = (idx & 2u) ? v67 : v45;
      return (idx & 4u) ? v47 : v03;
    }
    if (hc_lds()) return wuni(hc[(node * 8u + buf * 4u + a) << hsh]);  // (`hc` is a generic pointer: a flat load counts as a source of divergence)
    return nfm(node, HCO() + buf * NN() + a);
  }
  LBFT_HD void hc_set(u32 node, u32 buf, u32 a, u32 v) const {
    if (hc_reg()) {
      u32 idx = buf * 4u + a;
      LBFT_UNROLL
      for (u32 k = 0; k < 8; k++) hcw[k] = idx == k ? v : hcw[k];
      hcdirty = 1;
      return;
    }
    if (hc_lds()) hc[(node * 8u + buf * 4u + a) << hsh] = v;
    else nfms(node, HCO() + buf * NN() + a, v);
  }
  LBFT_HD void hcbr_to_lds() const {
    if (!hc_lds()) return;
    for (u32 node = 0; node < NN(); node++)
      for (u32 k = 0; k < 2 * NN(); k++) hc[(node * 8u + (k / NN()) * 4u + k % NN()) << hsh] = nfm(node, NF_FIXED_WORDS + k);
  }
  LBFT_HD void hcbr_from_lds() const {
    if (!hc_lds()) return;
    for (u32 node = 0; node < NN(); node++)
      for (u32 k = 0; k < 2 * NN(); k++) nfms(node, NF_FIXED_WORDS + k, hc[(node * 8u + (k / NN()) * 4u + k % NN()) << hsh]);
  }
  LBFT_HD void attach_tables(const u64* zx, const u64* zf, const u64* et) { zig_x = zx; zig_f = zf; exp_tab = et; }
  LBFT_HD void attach_peer_list(u8* list) { plist_lds = list; }
  // Large networks (33-128 nodes; the kernels that fit ONE cached block record in their registers): a direct-mapped window of hot block
  // records in LDS behind the register cache -- entry b mod N of this network's lane-private column: [N tags][N x BC_WORDS words].  The
  // 4-node kernel did not gain from such a window (its register cache misses 3 times per run); a 64-node network misses its single
  // register record on 75 % of 117 k lookups per run, each a dependent memory round trip (round 4, host-model counters).  Write-through
  // like the register cache (blk_put updates a resident entry), rebuilt empty at every launch.
  static constexpr bool BLW = CLS == K_LARGE_EXCHANGE;  // (compiled into lbft_k_run2q only: in lbft_k_run2l it cost 60 spilled registers and time, see lbft_hip.hip)
  u32* bl;       // nullptr = none
  u32 bl_n, bl_sh;
  LBFT_HD void attach_blk_window(u32* column, u32 entries, u32 stride_shift) { bl = (BLW && entries) ? column : nullptr; bl_n = entries; bl_sh = stride_shift; }
  LBFT_HD u32 blx(u32 k) const { return k << bl_sh; }
  LBFT_HD void blw_reset() const { if (BLW && bl) for (u32 e = 0; e < bl_n; e++) bl[blx(e)] = 0; }
  const u32* wtab;  // voting rights (the device attaches an LDS copy: weight() sits inside the vote / timeout insertion loops)
  LBFT_HD void attach_weights(const u32* w) { wtab = w; }
  LBFT_HD void attach_round_tables(const u8* leaders, u32 nl, const i64* durs, u32 nd) { leader_lds = leaders; leader_lds_len = nl; dur_lds = durs; dur_lds_len = nd; }

  LBFT_HD u32 rsh() const { return TILE64 ? 8u : IMAJOR ? 2u : P.rsh; }  // log2(row bytes)
  // (a multiplication, not `w << P.rsh`: hipcc 7.2 dies on the variable shift in this address pattern -- "Illegal instruction
  // detected: V_CMP_NE_U32 0, $src_shared_base" --; rows and row bytes are below 2^24, a plain 32-bit multiply is what compiles)
  LBFT_HD static u32 mul24(u32 a, u32 b) { return a * b; }  // (__umul24 trips the same compiler bug)
  LBFT_HD u32 rowb() const { return TILE64 ? 256u : IMAJOR ? 4u : 4u * P.tw; }  // bytes of a row
  LBFT_HD u32 boff(u32 w) const { return TILE64 ? (w << 8) + lane4 : IMAJOR ? (w << 2) + lane4 : mul24(w, rowb()) + lane4; }  // tile-relative byte offset of row w (a tile is < 4 GiB)
  LBFT_HD u32 ld(u32 w) const { LBFT_HOOK_MEM(boff(w), 0); return *reinterpret_cast<const u32*>(tile + (size_t)boff(w)); }
  LBFT_HD void st(u32 w, u32 v) const { LBFT_HOOK_MEM(boff(w), 1); *reinterpret_cast<u32*>(tile + (size_t)boff(w)) = v; }
  // row (w0 + f) given boff(w0): groups of 16 rows share one 32-bit base, the rest is the instruction's immediate
  LBFT_HD u32 ldf(u32 base, u32 f) const {
    LBFT_HOOK_MEM(IMAJOR ? base + f * 4u : base, 0);
    if (TILE64) return *reinterpret_cast<const u32*>(tile + (size_t)(base + (f & ~15u) * LBFT_ROW_BYTES) + (f & 15u) * LBFT_ROW_BYTES);
    if (IMAJOR) return *reinterpret_cast<const u32*>(tile + (size_t)base + f * 4u);  // (consecutive fields: immediate offsets, wide loads)
    return *reinterpret_cast<const u32*>(tile + (size_t)(base + mul24(f, rowb())));
  }
  LBFT_HD void stf(u32 base, u32 f, u32 v) const {
    LBFT_HOOK_MEM(IMAJOR ? base + f * 4u : base, 1);
    if (TILE64) *reinterpret_cast<u32*>(tile + (size_t)(base + (f & ~15u) * LBFT_ROW_BYTES) + (f & 15u) * LBFT_ROW_BYTES) = v;
    else if (IMAJOR) *reinterpret_cast<u32*>(tile + (size_t)base + f * 4u) = v;
    else *reinterpret_cast<u32*>(tile + (size_t)(base + mul24(f, rowb()))) = v;
  }

  // ---- field accessors ----
  LBFT_HD u32 nfw(u32 node, u32 f) const { return OFFNODE() + node * NWORDS() + f; }
  // direct (memory) access to any node's rows
  LBFT_HD u32 nfm(u32 node, u32 f) const { return ld(nfw(node, f)); }
  LBFT_HD void nfms(u32 node, u32 f, u32 v) const { st(nfw(node, f), v); }
  // Every event touches exactly one node.  Its fixed rows are staged in registers by begin_node()
  // (one burst of independent loads instead of a chain of dependent ones: the compiler cannot
  // reorder row loads across row stores) and the modified ones are written back by end_node().
  // `f` is a compile-time constant at every call site, so cw[] lives in VGPRs.
  mutable u32 cw[NF_FIXED_WORDS];
  LBFT_HD u32 cwg(u32 f) const { return cw[f]; }
  LBFT_HD void cws(u32 f, u32 v) const { cw[f] = v; }
  // rows are written back by groups of fields that change together: 6 tests instead of 41 (A/B on the 65536 x 4 batch in one GPU call:
  // 24.6 ms vs 25.1 ms per row; writing all rows unconditionally had measured 9 % slower).  cdirty holds one bit per GROUP.
  static constexpr u32 NGROUPS = 6;
  static constexpr u64 group_mask(u32 g) {
    return g == 0 ? (1ULL << NF_IGNORE_UNTIL) | (1ULL << NF_LAST_TIMER_T) | (1ULL << NF_TIMER_DUPS) | (1ULL << NF_DUP_STAMP)
         : g == 1 ? (1ULL << NF_PROPOSED_BLK) | (1ULL << NF_CUR_ROUND) | (1ULL << NF_TO_MASK) | (1ULL << NF_TO_WEIGHT) | (1ULL << NF_ELECTION) |
                    (1ULL << NF_BAL0_BLK) | (1ULL << NF_BAL0_WEIGHT) | (1ULL << NF_BAL0_AUTHORS) | (1ULL << NF_BAL1_BLK) | (1ULL << NF_BAL1_WEIGHT) | (1ULL << NF_BAL1_AUTHORS)
         : g == 2 ? (1ULL << NF_HQC_ROUND) | (1ULL << NF_HQC_BLK) | (1ULL << NF_HTC_ROUND) | (1ULL << NF_HC_ROUND) | (1ULL << NF_HCC_BLK) | (1ULL << NF_TC_MASK) | (1ULL << NF_TC_SEL)
         : g == 3 ? (1ULL << NF_PM_EPOCH) | (1ULL << NF_PM_ROUND) | (1ULL << NF_PM_LEADER) | (1ULL << NF_PM_START) | (1ULL << NF_PM_DUR_LO) | (1ULL << NF_PM_DUR_HI)
         : g == 4 ? (1ULL << NF_LVR) | (1ULL << NF_LOCKED) | (1ULL << NF_LQAT) | (1ULL << NF_TR_EPOCH) | (1ULL << NF_TR_HCR) | (1ULL << NF_TR_LCT) |
                    (1ULL << NF_NEXT_CMD) | (1ULL << NF_LAST_COMMITTED_BLK) | (1ULL << NF_NCOMMITS)
                  : (1ULL << NF_STARTUP) | (1ULL << NF_EPOCH) | (1ULL << NF_INIT_STATE_BLK) | (1ULL << NF_PREV_EPOCH_HCC);
  }
  static constexpr u32 group_of(u32 f) {
    return (group_mask(0) >> f) & 1 ? 0u : (group_mask(1) >> f) & 1 ? 1u : (group_mask(2) >> f) & 1 ? 2u : (group_mask(3) >> f) & 1 ? 3u : (group_mask(4) >> f) & 1 ? 4u : 5u;
  }
  static_assert(NF_FIXED_WORDS == 41 && (group_mask(0) | group_mask(1) | group_mask(2) | group_mask(3) | group_mask(4) | group_mask(5)) == (1ULL << 41) - 1,
                "every fixed word belongs to a write-back group");
  mutable u32 cdirty;
  LBFT_HD u32 nf(u32 node, u32 f) const { return f < NF_FIXED_WORDS ? cwg(f) : ld(nfw(node, f)); }
  LBFT_HD void nfs(u32 node, u32 f, u32 v) const {
    if (f < NF_FIXED_WORDS) { cws(f, v); cdirty |= 1u << group_of(f); }
    else st(nfw(node, f), v);
  }
  LBFT_HD void begin_node(u32 node) const {
    // one base pointer, then constant row offsets: the 38 loads become one burst with immediate offsets
    u32 nb = boff(OFFNODE() + node * NWORDS());
    LBFT_UNROLL
    for (u32 f = 0; f < NF_FIXED_WORDS; f++) cws(f, ldf(nb, f));
    cdirty = 0;
    hc_load(nb);
    ax_load(node);
  }
  LBFT_HD void end_node(u32 node) const {
    ax_store(node);
    u32 nb = boff(OFFNODE() + node * NWORDS());
    hc_store(nb);
    LBFT_UNROLL
    for (u32 g = 0; g < NGROUPS; g++)
      if ((cdirty >> g) & 1u) {
        LBFT_UNROLL
        for (u32 f = 0; f < NF_FIXED_WORDS; f++)
          if ((group_mask(g) >> f) & 1ULL) stf(nb, f, cwg(f));
      }
  }
  LBFT_HD u32 bfw(u32 b, u32 f) const { return OFFBLK() + (b - 1) * BWORDS() + f; }
  // (through the record's base + a field offset, NOT ld(bfw(b, f)): with instance-major rows `(first word + f) * 4 + lane offset` is
  // reassociated by the compiler into `first word * 4 + (f * 4 + lane offset)` and the bracket hoisted out of the event loop -- one
  // VGPR per distinct field for the whole launch (17 of them in lbft_k_run0q) and an add per access; with the base zero-extended first
  // the field offset becomes the instruction's immediate and consecutive fields merge into wide accesses)
  LBFT_HD u32 bf(u32 b, u32 f) const { return ldf(boff(bfw(b, 0)), f); }   // cold fields (B_TIME, B_CMD) and read-back
  LBFT_HD void bfs(u32 b, u32 f, u32 v) const { stf(boff(bfw(b, 0)), f, v); }
  LBFT_HD u32 blk_author(u32 b) const { return bf(b, B_LINK) >> 16; }

  // ---- block records: a FIFO of LBFT_BLK_CACHE hot records in registers.  Only this lane ever touches its
  // instance's block rows and every update is written through, so the cache stays valid for the whole
  // launch; the handful of blocks a network is working on (the proposal, its parent and grandparent, the
  // commit certificate) stop being memory round trips.  All indices below are compile-time after
  // unrolling, so the records live in VGPRs.
  struct Blk {
    u32 w[BC_WORDS];
    // n > 32: word xk (1..3) of the block's KNOWN / QC / PEND node sets, fetched together with the record for the node of the
    // current event (begin_node: cur_xk = node >> 5) -- the mask tests and updates of insert_block / insert_qc /
    // compute_state for nodes >= 32 then cost no memory round trip each.  xk = 0: not fetched (bm_* then go to memory).
    mutable u32 x[3], xk;
    LBFT_HD u32 round() const { return w[B_ROUND]; }
    LBFT_HD u32 prev() const { return w[B_LINK] & 0xffffu; }
    LBFT_HD u32 author() const { return w[B_LINK] >> 16; }
    LBFT_HD u32 prev_round() const { return w[B_PREV_ROUND]; }
    LBFT_HD u32 pp() const { return w[B_PP] & 0xffffu; }
    LBFT_HD u32 ppp() const { return w[B_PP] >> 16; }
    LBFT_HD u32 pp_round() const { return w[B_PP_ROUND]; }
    LBFT_HD u32 epoch() const { return w[B_EPOCH]; }
    LBFT_HD u32 depth() const { return w[B_DEPTH]; }
  };
#ifndef LBFT_BLK_CACHE_UNI
#define LBFT_BLK_CACHE_UNI 1  // lbft_k_run0u: its state lives in SGPRs / VGPR lanes (one v_readlane / v_writelane per access to a parked word), so every cached
                              // record is paid at every access to the state behind it -- round 5, 1 024 x 4: 3 records 7.56 ms, 2: 7.33, 1: 4.88 (lbft_k_run0s: 5.27)
#endif
  static constexpr u32 BCN = CLS == K_HEADLINE ? LBFT_BLK_CACHE_QUAD : CLS == K_SMALL_UNIFORM ? LBFT_BLK_CACHE_UNI : CLS == K_LARGE_LEAN ? LBFT_BLK_CACHE_LEAN5 : LEAN2 ? LBFT_BLK_CACHE_LEAN2 : LBFT_BLK_CACHE;
  mutable u32 bc_id[BCN];
  mutable u32 bc_w[BCN][BC_WORDS];
  mutable u32 bc_next;  // FIFO hand (plain round-robin replacement: the second-chance bookkeeping cost more than the misses it saved, EXPERIMENTS.md)
#if !defined(LBFT_NO_BC_PAD)
  mutable u32 bc_ref;   // (unused since the second-chance variant went; kept because WITHOUT this member the compiler emits 5 % more code for the
                        // large-network kernels -- lbft_k_run2l 135 -> 142 KB -- which round 5 measured: see EXPERIMENTS.md)
#endif
  LBFT_HD void blk_cache_reset() const {
    for (u32 e = 0; e < BCN; e++) bc_id[e] = 0;
    bc_next = 0;
#if !defined(LBFT_NO_BC_PAD)
    bc_ref = 0;
#endif
  }
  LBFT_HD void blk_cache_insert(u32 b, const Blk& r) const {
    LBFT_UNROLL
    for (u32 e = 0; e < BCN; e++) {
      // value selects at fixed entries, NOT `if (hand == e) entry[e] = r`: the compiler sinks such conditional stores
      // into one store through a phi of entry addresses, and an array addressed that way is no longer promoted to
      // registers (the whole cache ended up in scratch memory: 32 scratch loads per lookup)
      bool take = bc_next == e;
      bc_id[e] = take ? b : bc_id[e];
      LBFT_UNROLL
      for (u32 f = 0; f < BC_WORDS; f++) bc_w[e][f] = take ? r.w[f] : bc_w[e][f];
    }
    bc_next = bc_next + 1 == BCN ? 0 : bc_next + 1;
  }
  LBFT_HD void blw_fill(u32 b, const Blk& r) const {
    if (!BLW || !bl) return;
    const u32 e = b & (bl_n - 1u);
    bl[blx(e)] = b;
    LBFT_UNROLL
    for (u32 f = 0; f < BC_WORDS; f++) bl[blx(bl_n + e * BC_WORDS + f)] = r.w[f];
  }
  mutable u32 cur_xk;  // extension word of the node sets that the current event's node lives in (0: node < 32 or n <= 32)
  LBFT_HD Blk blk_get(u32 b) const {  // b != 0
    Blk r;
    bool hit = false;
    r.xk = 0; r.x[0] = r.x[1] = r.x[2] = 0;  // (node-set extension words: fetched by the first bm_* operation that needs them)
    // one select per further entry (the value of a miss is overwritten by the loads below): r = entry 0, then entry e where it is the hit
    bool he[BCN];
    LBFT_UNROLL
    for (u32 e = 0; e < BCN; e++) {
      he[e] = bc_id[e] == b;
      hit = hit || he[e];
    }
    LBFT_UNROLL
    for (u32 f = 0; f < BC_WORDS; f++) {
      u32 v = bc_w[0][f];
      LBFT_UNROLL
      for (u32 e = 1; e < BCN; e++) v = he[e] ? bc_w[e][f] : v;
      r.w[f] = v;
    }
    LBFT_COUNT(26);
    LBFT_STAT(44);
    if (!hit) {
      LBFT_COUNT(25);
      LBFT_STAT(45);
      LBFT_MARK(28);  // (diagnostic builds: the time since the previous mark, so that 29 is the miss alone)
      bool in_window = false;
      if (BLW && bl) {  // the LDS window first
        const u32 e = b & (bl_n - 1u);
        if (bl[blx(e)] == b) {
          in_window = true;
          LBFT_UNROLL
          for (u32 f = 0; f < BC_WORDS; f++) r.w[f] = bl[blx(bl_n + e * BC_WORDS + f)];
        }
      }
      if (!in_window) {
        u32 bb = boff(bfw(b, 0));
        LBFT_UNROLL
        for (u32 f = 0; f < BC_WORDS; f++) r.w[f] = ldf(bb, f);  // one burst of independent loads
        LBFT_DRAIN_VMEM();
        blw_fill(b, r);
      }
      LBFT_MARK(29);
      blk_cache_insert(b, r);
    }
    return r;
  }
  // Node sets of a block (B_KNOWN / B_QC / B_PEND): nodes 0..31 live in the hot record, nodes >= 32 (n > 32
  // only) in extension rows behind the cold fields.
  LBFT_HD u32 bxw(u32 b, u32 f, u32 k) const { return bfw(b, B_WORDS + (f - B_KNOWN) * (MW() - 1) + k - 1); }
  // (one round trip for the three words of this record copy instead of one per test / update)
  LBFT_HD void bx_fetch(u32 b, const Blk& rb, u32 k) const {
    if (rb.xk == k) return;
    rb.x[0] = ld(bxw(b, B_KNOWN, k)); rb.x[1] = ld(bxw(b, B_QC, k)); rb.x[2] = ld(bxw(b, B_PEND, k));
    rb.xk = k;
  }
  LBFT_HD bool bm_test(u32 b, const Blk& rb, u32 f, u32 node) const {
    if (!wide() || node < 32) return (rb.w[f] >> node) & 1u;
    if (F_BX) {
      bx_fetch(b, rb, node >> 5);
      return (rb.x[f - B_KNOWN] >> (node & 31u)) & 1u;
    }
    return (ld(bxw(b, f, node >> 5)) >> (node & 31u)) & 1u;
  }
  // does `node` hold block b AND its quorum certificate?  (read from the block's row, not through the register cache: write-through keeps the rows current)
  LBFT_HD bool bm_known_qc(u32 b, u32 node) const {
    u32 kw, qw;
    if (!wide() || node < 32) { const u32 bb = boff(bfw(b, 0)); kw = ldf(bb, B_KNOWN); qw = ldf(bb, B_QC); }
    else { kw = ld(bxw(b, B_KNOWN, node >> 5)); qw = ld(bxw(b, B_QC, node >> 5)); }
    return ((kw & qw) >> (node & 31u)) & 1u;
  }
  LBFT_HD void bm_set(u32 b, Blk& rb, u32 f, u32 node) const {
    if (!wide() || node < 32) { rb.w[f] |= 1u << node; blk_put(b, f, rb.w[f]); return; }
    if (F_BX) {
      bx_fetch(b, rb, node >> 5);
      rb.x[f - B_KNOWN] |= 1u << (node & 31u);
      st(bxw(b, f, node >> 5), rb.x[f - B_KNOWN]);
    } else { u32 w = bxw(b, f, node >> 5); st(w, ld(w) | (1u << (node & 31u))); }
  }
  LBFT_HD void bm_clr(u32 b, Blk& rb, u32 f, u32 node) const {
    if (!wide() || node < 32) { rb.w[f] &= ~(1u << node); blk_put(b, f, rb.w[f]); return; }
    if (F_BX) {
      bx_fetch(b, rb, node >> 5);
      rb.x[f - B_KNOWN] &= ~(1u << (node & 31u));
      st(bxw(b, f, node >> 5), rb.x[f - B_KNOWN]);
    } else { u32 w = bxw(b, f, node >> 5); st(w, ld(w) & ~(1u << (node & 31u))); }
  }
  // Write-through update of one mask word of block b (f is B_KNOWN, B_QC or B_PEND).
  LBFT_HD void blk_put(u32 b, u32 f, u32 v) const {
    bfs(b, f, v);
    if (BLW && bl) {
      const u32 e = b & (bl_n - 1u);
      if (bl[blx(e)] == b) bl[blx(bl_n + e * BC_WORDS + f)] = v;
    }
    LBFT_UNROLL
    for (u32 e = 0; e < BCN; e++) {  // (value selects: see blk_cache_insert)
      bool hit = bc_id[e] == b;
      if (f == B_KNOWN) bc_w[e][B_KNOWN] = hit ? v : bc_w[e][B_KNOWN];
      else if (f == B_QC) bc_w[e][B_QC] = hit ? v : bc_w[e][B_QC];
      else bc_w[e][B_PEND] = hit ? v : bc_w[e][B_PEND];
    }
  }
  LBFT_HD u32 sfw(u32 slot, u32 f) const { return OFFSNAP() + slot * SWORDS() + f; }
  // extension word k >= 1 of a snapshot's TC (which = 0) / current-timeout (which = 1) author set
  LBFT_HD u32 sxw(u32 slot, u32 which, u32 k) const { return sfw(slot, S_FIXED_WORDS + 2 * NN() + which * (MW() - 1) + k - 1); }

  // instance scalars: row `f` of the instance through its first row + an immediate (see bf(): a plain ld(I_X) costs a VGPR per scalar
  // that stays live from the loads at the start of a launch to the stores at its end)
  LBFT_HD u32 ldi(u32 f) const { return ldf(boff(0), f); }
  LBFT_HD void sti(u32 f, u32 v) const { stf(boff(0), f, v); }
  LBFT_HD void load_scalars() {
    clock = (i32)ldi(I_CLOCK); stamp = ldi(I_STAMP);
    rng.s0 = ldi(I_RNG0) | ((u64)ldi(I_RNG1) << 32); rng.s1 = ldi(I_RNG2) | ((u64)ldi(I_RNG3) << 32);
    rng.s2 = ldi(I_RNG4) | ((u64)ldi(I_RNG5) << 32); rng.s3 = ldi(I_RNG6) | ((u64)ldi(I_RNG7) << 32);
    rng.draws = ldi(I_DRAWS);
    qlen = ldi(I_QLEN); snap_free = ldi(I_SNAP_FREE); nblocks = ldi(I_NBLOCKS); fault = ldi(I_FAULT);
    ev0 = ldi(I_EV0); ev1 = ldi(I_EV1); ev2 = ldi(I_EV2); ev3 = ldi(I_EV3);
    maxq = ldi(I_MAXQ); maxsnap = ldi(I_MAXSNAP);
    if (!LEAN2) {  // (state the two-wavefront large-network kernels never touch stays in its rows)
      snap_mask = ldi(I_SNAP_MASK_LO) | ((u64)ldi(I_SNAP_MASK_HI) << 32);
      last_node = ldi(I_LAST_NODE); vd_time = ldi(I_VD_TIME); vd_stamp = ldi(I_VD_STAMP);
    } else { snap_mask = 0; last_node = 0; vd_time = 0xffffffffu; vd_stamp = 0; }
    cal_cursor = ldi(I_CAL_CURSOR); cal_free = ldi(I_CAL_FREE); cal_bump = ldi(I_CAL_BUMP);
    n_fold = ldi(I_NFOLD); n_upd = ldi(I_NUPD);
    cont = ldi(I_CONT);
    sp_idx = 0; sp_s1 = 0; sp_meta = 0; sp_nx = 0; cur_h = 0;
    if (RING) { rng.rhead = ldi(I_RING_HEAD); rng.rcnt = ldi(I_RING_CNT); }
    blk_cache_reset();
  }
  LBFT_HD void store_scalars(bool done) {
    sti(I_CLOCK, (u32)clock); sti(I_STAMP, stamp);
    sti(I_RNG0, (u32)rng.s0); sti(I_RNG1, (u32)(rng.s0 >> 32)); sti(I_RNG2, (u32)rng.s1); sti(I_RNG3, (u32)(rng.s1 >> 32));
    sti(I_RNG4, (u32)rng.s2); sti(I_RNG5, (u32)(rng.s2 >> 32)); sti(I_RNG6, (u32)rng.s3); sti(I_RNG7, (u32)(rng.s3 >> 32));
    sti(I_DRAWS, rng.draws);
    sti(I_QLEN, qlen); sti(I_SNAP_FREE, snap_free); sti(I_NBLOCKS, nblocks); sti(I_FAULT, fault);
    sti(I_EV0, ev0); sti(I_EV1, ev1); sti(I_EV2, ev2); sti(I_EV3, ev3);
    sti(I_MAXQ, maxq); sti(I_MAXSNAP, maxsnap);
    if (!LEAN2) {
      sti(I_SNAP_MASK_LO, (u32)snap_mask); sti(I_SNAP_MASK_HI, (u32)(snap_mask >> 32));
      sti(I_LAST_NODE, last_node); sti(I_VD_TIME, vd_time); sti(I_VD_STAMP, vd_stamp);
    }
    if (cal() && cur_h) { st(calh(sp_idx), cur_h); cur_h = 0; sp_s1 = 0; }  // (the open bucket's head lives in a register while it drains)
    sti(I_CAL_CURSOR, cal_cursor); sti(I_CAL_FREE, cal_free); sti(I_CAL_BUMP, cal_bump);
    sti(I_NFOLD, n_fold); sti(I_NUPD, n_upd);
    if (q1()) sti(I_CONT, cont);
    if (RING) { sti(I_RING_HEAD, rng.rhead); sti(I_RING_CNT, rng.rcnt); }
    sti(I_DONE, done ? 1u : 0u);
  }

  // ---- network delay (simulator.rs:110-118; rand_distr 0.4 LogNormal / StandardNormal ziggurat) ----
  LBFT_HD double standard_normal() {
    const double R = lbft_asdouble(0x400d3bb48209ad33ULL);  // ZIG_NORM_R
    for (;;) {
      u64 bits = rng.next_u64();
      u32 i = (u32)(bits & 0xff);
      double u = lbft_asdouble((1024ULL << 52) | (bits >> 12)) - 3.0;
      double xi = lbft_asdouble(zig_x[i]);
      double x = u * xi;
      double ax = x < 0.0 ? -x : x;
      if (LBFT_LIKELY(ax < lbft_asdouble(zig_x[i + 1]))) return x;
      if (LBFT_UNLIKELY(i == 0)) {
        double xx = 1.0, yy = 0.0;
        while (-2.0 * yy < xx * xx) {
          double a = lbft_asdouble((1023ULL << 52) | (rng.next_u64() >> 12)) - (1.0 - 0x1p-53);
          double b = lbft_asdouble((1023ULL << 52) | (rng.next_u64() >> 12)) - (1.0 - 0x1p-53);
          xx = wuni(lbft_log(a)) / R;
          yy = wuni(lbft_log(b));
        }
        return u < 0.0 ? xx - R : R - xx;
      }
      double f0 = lbft_asdouble(zig_f[i]), f1 = lbft_asdouble(zig_f[i + 1]);
      double f01 = (double)(rng.next_u64() >> 11) * 0x1p-53;
      if (f1 + (f0 - f1) * f01 < wuni(lbft_exp(-x * x / 2.0, exp_tab))) return x;
    }
  }
  // (i64) exp(y) as the reference truncates it (simulator.rs:115-117), without evaluating the exact exp when a cheap estimate
  // already decides the integer: a = 2^(float)(y log2 e) from the hardware's single-precision exp2 is within 8e-7 (relative) of
  // exp(y) for |y log2 e| < 20 -- 6.6e-7 from rounding the argument to single precision (half an ulp of 2^-19, times ln 2),
  // 1.2e-7 from the instruction (1 ulp), the double-precision product is exact to 1e-15 -- and lbft_exp (= glibc's exp) is within
  // 1e-16 of it.  So whenever a lies further than 2e-6 a from both neighbouring integers, trunc(exp(y)) = floor(a); otherwise
  // (one sample in ~10^4) the exact routine decides.  The result is the reference's in every case; only the work differs.
  LBFT_HD i64 trunc_exp(double y) const {
#if LBFT_FAST_TRUNC_EXP
    float t = (float)(y * 1.4426950408889634);
    if (LBFT_LIKELY(t > -20.0f && t < 20.0f)) {
#if defined(__HIP_DEVICE_COMPILE__)
      float a = __builtin_amdgcn_exp2f(t);
#else
      float a = exp2f(t);
#endif
      float fl = __builtin_floorf(a), fr = a - fl, m = a * 2e-6f;
      if (LBFT_LIKELY(fr > m && fr < 1.0f - m)) return (i64)(i32)fl;
    }
#endif
    return f64_to_i64_sat(wuni(lbft_exp(y, exp_tab)));
  }
  LBFT_HD i64 sample_delay() {
    if (DMODEL() == 1) return P.uni_lo + (i64)rng.gen_range_u64(P.uni_span);
    double nrm = standard_normal();
    return trunc_exp(P.mu + P.sigma * nrm);
  }

  // ---- event queue: unsorted compact array, ordered by (time asc, kind desc, stamp asc)
  //      (ScheduledEvent::cmp, simulator.rs:149-161).  Events scheduled after max_clock can never
  //      run (loop_until breaks at the first one, simulator.rs:389) and are dropped at push time;
  //      they still consume a creation stamp.
  // Packed entries (kernel class 0: n <= 16, scap <= 256, max_clock < 2^21): one 64-bit word per event,
  //   [63:43] time  [42:41] 3 - kind  [40:16] creation stamp  [15:12] node  [11:8] sender  [7:0] snapshot slot
  // Stamps are unique, so comparing whole words orders events exactly like (time, 3 - kind, stamp).  The LDS front then
  // holds keys only (8 instead of 12 bytes per slot) and a pop reads one word per slot.  Unused LDS slots hold the
  // sentinel ~0 (larger than any entry), so the scan needs no per-slot bound check and runs in batches of 8 loads.
#define LBFT_QP_TIME_BITS 21
#define LBFT_QP_STAMP_BITS 25
  LBFT_HD void q_set(u32 k, u64 key, u32 meta) const {
    if (qpacked()) {
      if (LBFT_LIKELY(k < ql)) qk[qx(k)] = key;
      else { st(P.off_qhi + k, (u32)(key >> 32)); st(P.off_qlo + k, (u32)key); }
      return;
    }
    if (k < ql) { qk[qx(k)] = key; qm[qx(k)] = meta; }
    else { st(P.off_qhi + k, (u32)(key >> 32)); st(P.off_qlo + k, (u32)key); st(P.off_qmeta + k, meta); }
  }
  LBFT_HD void q_get(u32 k, u64& key, u32& meta) const {
    if (qpacked()) {
      meta = 0;
      if (LBFT_LIKELY(k < ql)) key = qk[qx(k)];
      else key = ((u64)ld(P.off_qhi + k) << 32) | ld(P.off_qlo + k);
      return;
    }
    if (k < ql) { key = qk[qx(k)]; meta = qm[qx(k)]; }
    else { key = ((u64)ld(P.off_qhi + k) << 32) | ld(P.off_qlo + k); meta = ld(P.off_qmeta + k); }
  }
  // The LDS front is a cache of the HBM rows between launches.
  LBFT_HD void queue_to_lds() const {
    u32 nl = qlen < ql ? qlen : ql;
    for (u32 k = 0; k < nl; k++) {
      qk[qx(k)] = ((u64)ld(P.off_qhi + k) << 32) | ld(P.off_qlo + k);
      if (!qpacked()) qm[qx(k)] = ld(P.off_qmeta + k);
    }
    if (qpacked()) for (u32 k = nl; k < ql; k++) qk[qx(k)] = ~0ULL;
  }
  LBFT_HD void queue_from_lds() const {
    u32 nl = qlen < ql ? qlen : ql;
    for (u32 k = 0; k < nl; k++) {
      u64 key = qk[qx(k)];
      st(P.off_qhi + k, (u32)(key >> 32)); st(P.off_qlo + k, (u32)key);
      if (!qpacked()) st(P.off_qmeta + k, qm[qx(k)]);
    }
  }
  // calendar queue rows: (head, tail) word pairs per bucket, chunk pool (rows off_qmeta), stack of freed chunks (rows off_qlo)
#define LBFT_CAL_CH 32u  // words per chunk
#define LBFT_CAL_CE 31u  // entries per chunk
  LBFT_HD u32 calh(u32 idx) const { return P.off_cal_head + 2u * idx; }       // (chunk + 1) << 6 | position of the next entry to pop
  LBFT_HD u32 calt(u32 idx) const { return P.off_cal_head + 2u * idx + 1u; }  // (chunk + 1) << 6 | position of the last entry written; 0 = empty bucket
  LBFT_HD u32 chw(u32 c, u32 pos) const { return P.off_qmeta + c * LBFT_CAL_CH + pos; }
  LBFT_HD u32 cal_alloc_chunk() {
    u32 c;
    if (cal_free) c = ld(P.off_qlo + --cal_free);
    else c = cal_bump++;
    if (LBFT_UNLIKELY(c >= P.cal_chunks)) { fault |= F_QUEUE_OVERFLOW; c = P.cal_chunks - 1u; }  // (in bounds; the instance's results are void from here on)
    return c;
  }
  // opens the first non-empty bucket at or after the cursor, unless that is the one already open (the queue is not empty)
  LBFT_HD void cal_open() {
    if (cur_h != 0 && cal_cursor == sp_idx) return;
    if (cur_h != 0) st(calh(sp_idx), cur_h);  // (a push lowered the cursor below the open bucket: park it)
    u32 w = cal_cursor >> 5;
    u32 raw = ld(P.off_cal_bm + w);
    u32 bits = raw & (~0u << (cal_cursor & 31u));
    while (!bits) { w++; raw = ld(P.off_cal_bm + w); bits = raw; }
    u32 idx = w * 32u + ctz32(bits);
    cal_cursor = idx; sp_idx = idx;
    cur_h = ld(calh(idx)); sp_nx = ld(calt(idx));
    sp_s1 = 0;
  }
  // `cnt` entries of the open bucket, from its head on and all in the head's chunk, have been consumed
  LBFT_HD void cal_advance(u32 cnt) {
    const u32 idx = sp_idx, c = (cur_h >> 6) - 1u, lastpos = (cur_h & 63u) + cnt - 1u;
    bool last = (((c + 1u) << 6) | lastpos) == sp_nx;
    if (last) {  // ... unless the bucket was appended to while it drained (a zero-delay send into the open bucket)
      u32 t2 = ld(calt(idx));
      if (t2 != sp_nx) { sp_nx = t2; last = false; }
    }
    if (last) {
      st(calh(idx), 0); st(calt(idx), 0);
      u32 bw = P.off_cal_bm + (idx >> 5);
      st(bw, ld(bw) & ~(1u << (idx & 31u)));
      st(P.off_qlo + cal_free++, c);  // stack of freed chunks
      cur_h = 0;
    } else if (lastpos == LBFT_CAL_CE) {
      u32 nx = ld(chw(c, 0));
      st(P.off_qlo + cal_free++, c);
      cur_h = (nx << 6) | 1u;
    } else cur_h = ((c + 1u) << 6) | (lastpos + 1u);
    sp_s1 = 0;
    if (F_SPEC && cur_h != 0) { sp_meta = ld(chw((cur_h >> 6) - 1u, cur_h & 63u)); sp_s1 = 1; }  // (not the last one: the next entry exists already)
  }
  // `reuse_stamp` != ~0u: the event takes that (already handed out, otherwise unused) creation stamp.
  LBFT_HD bool push_event(i64 time, u32 kind, u32 node, u32 sender, u32 slot, u32 reuse_stamp = ~0u) {
    u32 my_stamp = reuse_stamp;
    if (reuse_stamp == ~0u) my_stamp = stamp++;
    if (LBFT_UNLIKELY(time > (i64)P.max_clock)) return false;
    if (LBFT_UNLIKELY(my_stamp >= (qpacked() ? (1u << LBFT_QP_STAMP_BITS) : (1u << 30)))) { fault |= F_STAMP_OVERFLOW; return false; }
    if (LBFT_UNLIKELY(qlen >= P.qcap)) { fault |= F_QUEUE_OVERFLOW; return false; }
    LBFT_HOOK_PUSH(time, kind, node);
    u64 key = ((u64)(u32)time << 32) | ((3u - kind) << 30) | my_stamp;
    u32 meta = node | (sender << 8) | (slot << 16);
    if (qpacked())
      key = ((u64)(u32)time << 43) | ((u64)(3u - kind) << 41) | ((u64)my_stamp << 16) | (u64)((node << 12) | (sender << 8) | slot);
    if (cal()) {
      // Calendar queue: bucket = (time, kind) in pop order; creation stamps grow with every push, so appending
      // keeps each bucket sorted by stamp and the key never has to be stored or compared.  O(1), ~1 round trip.
      // Round 6: a bucket is a chain of CHUNKS of 31 consecutive entries (one 128-byte line: word 0 = next chunk + 1, words 1..31 = event metas)
      // instead of a linked list of single slots: an append is the tail word + one store into the tail chunk (the old form wrote the slot's meta,
      // its link, its predecessor's link and read the free-slot stack: four more lines), a pop reads consecutive words of a line it already has.
      u32 idx = (u32)time * 4u + (3u - kind);
      u32 tl = ld(calt(idx));
      u32 c, pos;
      if (tl == 0) {  // empty bucket
        c = cal_alloc_chunk(); pos = 1;
        st(calh(idx), ((c + 1u) << 6) | 1u);
        u32 bw = P.off_cal_bm + (idx >> 5);
        st(bw, ld(bw) | (1u << (idx & 31u)));
      } else if ((tl & 63u) == LBFT_CAL_CE) {  // the tail chunk is full
        c = cal_alloc_chunk(); pos = 1;
        st(chw((tl >> 6) - 1u, 0), c + 1u);
      } else { c = (tl >> 6) - 1u; pos = (tl & 63u) + 1u; }
      st(chw(c, pos), meta);
      st(calt(idx), ((c + 1u) << 6) | pos);
      if (idx < cal_cursor) cal_cursor = idx;
    } else if (heap()) {  // large networks: binary min-heap in the HBM rows, sift up
      u32 i = qlen;
      while (i > 0) {
        u32 par = (i - 1) >> 1;
        u64 pk; u32 pm;
        q_get(par, pk, pm);
        if (pk < key) break;
        q_set(i, pk, pm);
        i = par;
      }
      q_set(i, key, meta);
    } else {
      q_set(qlen, key, meta);
    }
    qlen++;
    if (qlen > maxq) maxq = qlen;
    return true;
  }
  // smallest of kk[LO .. LO + N) and its index: a tree of compare-selects over values (constant indices at the leaves)
  template <u32 LO, u32 N> LBFT_HD static void qmin(const u64* kk, u64& m, u32& i) {
    if constexpr (N == 1) { m = kk[LO]; i = LO; }
    else {
      u64 ma, mb; u32 ia, ib;
      qmin<LO, N / 2>(kk, ma, ia);
      qmin<LO + N / 2, N - N / 2>(kk, mb, ib);
      bool lt = mb < ma;
      m = lt ? mb : ma; i = lt ? ib : ia;
    }
  }
  // Packed queue: (bkey, best) = the smallest key among the LDS slots and its slot.  Completes the pop: the spilled tail (slots >= ql
  // live in the HBM rows; rare when ql covers the high-water mark) is compared, the event decoded, and the last entry moved into the hole.
  LBFT_HD void pop_take(u64 bkey, u32 best, i32& time, u32& kind, u32& meta) {
    if (LBFT_UNLIKELY(qlen > ql))
      for (u32 k = ql; k < qlen; k++) {
        u64 key = ((u64)ld(P.off_qhi + k) << 32) | ld(P.off_qlo + k);
        if (key < bkey) { bkey = key; best = k; }
      }
    time = (i32)(u32)(bkey >> 43);
    kind = 3u - ((u32)(bkey >> 41) & 3u);
    ev_stamp = (u32)(bkey >> 16) & ((1u << LBFT_QP_STAMP_BITS) - 1u);
    u32 lo = (u32)bkey;
    meta = ((lo >> 12) & 15u) | (((lo >> 8) & 15u) << 8) | ((lo & 0xffu) << 16);
    qlen--;
    u64 lk = ~0ULL; u32 lm = 0;
    if (best != qlen) { q_get(qlen, lk, lm); q_set(best, lk, 0); }
    if (qlen < ql) qk[qx(qlen)] = ~0ULL;  // the vacated last slot becomes a sentinel again
  }
  // ---- the pop's scan by ALL 64 lanes of the wavefront (kernel class 0, SimT::run_popc; the north star's "next-event-time reductions
  // done with wavefront shuffle primitives").  A wavefront carries lpw networks in its first lanes; their LDS queue columns are one
  // array keys[slot][lpw] (unused slots hold the sentinel ~0), so lane l reads the words l, l + 64, l + 128, ... of that array -- slot
  // (l + 64 j) / lpw of network l % lpw: every network's slots spread over 64 / lpw lanes, the idle lanes of a small-batch wavefront (63
  // of 64 at one network per wavefront) and the upper half of a 32-network wavefront included.  Each lane keeps the smallest of its
  // ql lpw / 64 keys, a butterfly of wavefront shuffles over the lanes of equal l % lpw leaves every network's minimum in all of its
  // lanes, and -- keys are unique (creation stamps) -- the lane holding it names the slot.  `kw`: the wavefront's key array.
  LBFT_HD void coop_find(const u64* kw, u64& bkey, u32& best) const {
#if defined(__HIP_DEVICE_COMPILE__)
    const u32 lane = lbft_lane_id();
    const u32 words = ql << qsh;  // ql * lpw
    u64 m = ~0ULL; u32 mi = 0;
    for (u32 w0 = 0; w0 < words; w0 += 256u) {  // four independent loads per round trip
      u64 v0, v1, v2, v3;
      if (WUNI) {  // branch-free (clamped index + select): no lane-dependent branch inside the otherwise wavefront-uniform event loop
        const u32 last = words - 1u;
        const u32 i0 = w0 + lane, i1 = w0 + 64u + lane, i2 = w0 + 128u + lane, i3 = w0 + 192u + lane;
        v0 = kw[i0 < last ? i0 : last]; v1 = kw[i1 < last ? i1 : last]; v2 = kw[i2 < last ? i2 : last]; v3 = kw[i3 < last ? i3 : last];
        v0 = i0 < words ? v0 : ~0ULL; v1 = i1 < words ? v1 : ~0ULL; v2 = i2 < words ? v2 : ~0ULL; v3 = i3 < words ? v3 : ~0ULL;
      } else {
        v0 = w0 + lane < words ? kw[w0 + lane] : ~0ULL; v1 = w0 + 64u + lane < words ? kw[w0 + 64u + lane] : ~0ULL;
        v2 = w0 + 128u + lane < words ? kw[w0 + 128u + lane] : ~0ULL; v3 = w0 + 192u + lane < words ? kw[w0 + 192u + lane] : ~0ULL;
      }
      bool a = v1 < v0, b = v3 < v2;
      u64 m01 = a ? v1 : v0, m23 = b ? v3 : v2;
      u32 i01 = a ? w0 + 64u + lane : w0 + lane, i23 = b ? w0 + 192u + lane : w0 + 128u + lane;
      bool c = m23 < m01;
      u64 mm = c ? m23 : m01; u32 ii = c ? i23 : i01;
      if (mm < m) { m = mm; mi = ii; }
    }
    if (WUNI || qstr == 1u) {  // (WUNI: at compile time -- the other exit's per-lane result would make every later value count as divergent)
      // ONE network in the wavefront: a DPP reduction (row shifts, then the two row broadcasts) leaves the minimum in lane 63 -- no LDS
      // crossbar round trips at all --, a scalar read hands it to every lane, and the lane holding it names the slot
      u32 lo = (u32)m, hi = (u32)(m >> 32);
#define LBFT_DPP_MIN_STEP(ctrl, rmask)                                                                                    \
      {                                                                                                                     \
        u32 olo = (u32)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)lo, ctrl, rmask, 0xf, false);                      \
        u32 ohi = (u32)__builtin_amdgcn_update_dpp((int)0xffffffffu, (int)hi, ctrl, rmask, 0xf, false);                      \
        bool lt = ohi < hi || (ohi == hi && olo < lo);                                                                      \
        lo = lt ? olo : lo; hi = lt ? ohi : hi;                                                                             \
      }
      LBFT_DPP_MIN_STEP(0x111, 0xf)  // row_shr:1
      LBFT_DPP_MIN_STEP(0x112, 0xf)  // row_shr:2
      LBFT_DPP_MIN_STEP(0x114, 0xf)  // row_shr:4
      LBFT_DPP_MIN_STEP(0x118, 0xf)  // row_shr:8   -> lane 15 of every row: the row's minimum
      LBFT_DPP_MIN_STEP(0x142, 0xa)  // row_bcast:15 into rows 1 and 3
      LBFT_DPP_MIN_STEP(0x143, 0xc)  // row_bcast:31 into rows 2 and 3 -> lane 63: the wavefront's minimum
#undef LBFT_DPP_MIN_STEP
      const u32 glo = (u32)__builtin_amdgcn_readlane((int)lo, 63), ghi = (u32)__builtin_amdgcn_readlane((int)hi, 63);
      const u64 g = ((u64)ghi << 32) | glo;
      const unsigned long long holders = __ballot(m == g && g != ~0ULL);
      const u32 hl = holders ? (u32)__builtin_ctzll(holders) : 0u;
      bkey = g; best = (u32)__builtin_amdgcn_readlane((int)mi, (int)hl);  // (qsh == 0: the word index is the slot)
      return;
    }
    for (u32 d = qstr; d < 64u; d <<= 1) {  // lanes l ^ d share l % lpw
      u32 olo = (u32)__shfl_xor((int)(u32)m, (int)d, 64), ohi = (u32)__shfl_xor((int)(u32)(m >> 32), (int)d, 64), oi = (u32)__shfl_xor((int)mi, (int)d, 64);
      u64 o = ((u64)ohi << 32) | olo;
      bool lt = o < m;
      m = lt ? o : m; mi = lt ? oi : mi;
    }
    bkey = m; best = mi >> qsh;
#else
    (void)kw; bkey = ~0ULL; best = 0;
#endif
  }
  // The same for wavefronts that carry many networks (lbft_k_run0q: 16 or 32): lane l belongs to column l % lpw and to scan group
  // l / lpw (64 / lpw groups); a column's batches of LBFT_POP_BATCH slots are dealt to its groups in turn and stop at the queue's length,
  // which the column's leader lane (lane l % lpw) holds in `qn` (0: no live network in that column).
  LBFT_HD void coop_find_cols(const u64* kw, u32 qn, u64& bkey, u32& best) const {
#if defined(__HIP_DEVICE_COMPILE__)
    const u32 lane = lbft_lane_id();
    const u32 col = lane & (QSTR() - 1u), grp = lane >> QSH(), groups = 64u >> QSH();
    const u32 qcol = (u32)__shfl((int)qn, (int)col, 64);
    const u32 nl = qcol < ql ? qcol : ql;
    u64 m = ~0ULL; u32 mi = 0;
    for (u32 k0 = grp * PB; k0 < nl; k0 += groups * PB) {
      u64 kk[PB];
LBFT_UNROLL
      for (u32 j = 0; j < PB; j++) kk[j] = kw[((k0 + j) << QSH()) + col];
      u64 bm; u32 bi;
      qmin<0, PB>(kk, bm, bi);
      if (bm < m) { m = bm; mi = k0 + bi; }
    }
    for (u32 d = QSTR(); d < 64u; d <<= 1) {
      u32 olo = (u32)__shfl_xor((int)(u32)m, (int)d, 64), ohi = (u32)__shfl_xor((int)(u32)(m >> 32), (int)d, 64), oi = (u32)__shfl_xor((int)mi, (int)d, 64);
      u64 o = ((u64)ohi << 32) | olo;
      bool lt = o < m;
      m = lt ? o : m; mi = lt ? oi : mi;
    }
    bkey = m; best = mi;
#else
    (void)kw; (void)qn; bkey = ~0ULL; best = 0;
#endif
  }
  // Removes the minimum; returns false when the queue is empty.
  LBFT_HD bool pop_event(i32& time, u32& kind, u32& meta) {
    if (qlen == 0) return false;
    LBFT_STAT(48 + (qlen > 56 ? 7 : qlen / 8));
    if (cal()) {  // first non-empty bucket at or after the cursor, head of its FIFO
      // The bucket being drained stays "open" in registers (sp_idx, cur_h = its head word, sp_nx = its tail word as last seen): a pop is ONE load of the
      // next entry -- whose address is known a step ahead, so it is fetched ahead (sp_meta) -- and no store at all; bitmap, head and tail words are only
      // touched when a bucket is opened or drained, the chunk's link word once per 31 entries.
      cal_open();
      const u32 idx = sp_idx;
      meta = (F_SPEC && sp_s1) ? sp_meta : ld(chw((cur_h >> 6) - 1u, cur_h & 63u));
      cal_advance(1);
      time = (i32)(idx >> 2);
      kind = 3u - (idx & 3u);
      ev_stamp = 0;                           // (only the round trace needs stamps; it runs on the heap queue)
      qlen--;
      return true;
    }
    if (heap()) {  // binary min-heap: take the root, sift the last entry down from the top
      u64 rk; u32 rm;
      q_get(0, rk, rm);
      time = (i32)(u32)(rk >> 32);
      kind = 3u - ((u32)rk >> 30);
      ev_stamp = (u32)rk & 0x3fffffffu;
      meta = rm;
      qlen--;
      if (qlen) {
        u64 lk; u32 lm;
        q_get(qlen, lk, lm);
        u32 i = 0;
        for (;;) {
          u32 c = 2 * i + 1;
          if (c >= qlen) break;
          u64 ck; u32 cm;
          q_get(c, ck, cm);
          if (c + 1 < qlen) {
            u64 dk; u32 dm;
            q_get(c + 1, dk, dm);
            if (dk < ck) { ck = dk; cm = dm; c = c + 1; }
          }
          if (lk < ck) break;
          q_set(i, ck, cm);
          i = c;
        }
        q_set(i, lk, lm);
      }
      return true;
    }
    u32 best = 0;
    u64 bkey = ~0ULL;
    u32 nl = qlen < ql ? qlen : ql;
    if (qpacked()) {
      // ql is a multiple of the batch and slots >= qlen hold the sentinel: LBFT_POP_BATCH independent loads in flight per batch,
      // then a tree of compare-selects (a sequential min pays one LDS round trip per slot)
      for (u32 k0 = 0; k0 < nl; k0 += PB) {
        u64 kk[PB];
        LBFT_UNROLL
        for (u32 j = 0; j < PB; j++) kk[j] = qk[qx(k0 + j)];
        u64 bm; u32 bi;
        qmin<0, PB>(kk, bm, bi);
        if (bm < bkey) { bkey = bm; best = k0 + bi; }
      }
      pop_take(bkey, best, time, kind, meta);
      return true;
    }
    LBFT_UNROLL4
    for (u32 k = 0; k < nl; k++) {
      u64 key = qk[qx(k)];
      if (key < bkey) { bkey = key; best = k; }
    }
    for (u32 k = ql; k < qlen; k++) {  // spilled tail (rare when ql covers the high-water mark)
      u64 key = ((u64)ld(P.off_qhi + k) << 32) | ld(P.off_qlo + k);
      if (key < bkey) { bkey = key; best = k; }
    }
    time = (i32)(u32)(bkey >> 32);
    kind = 3u - ((u32)bkey >> 30);
    ev_stamp = (u32)bkey & 0x3fffffffu;
    meta = best < ql ? qm[qx(best)] : ld(P.off_qmeta + best);
    qlen--;
    if (best != qlen) {
      u64 lk; u32 lm;
      q_get(qlen, lk, lm);
      q_set(best, lk, lm);
    }
    return true;
  }

  // ---- snapshot slots (free stack + refcount) ----
  LBFT_HD u32 ctz64(u64 x) const { return (u32)x ? ctz32((u32)x) : 32u + ctz32((u32)(x >> 32)); }
  LBFT_HD u32 popc64(u64 x) const {
#if defined(__HIP_DEVICE_COMPILE__)
    return (u32)__popcll(x);
#else
    return (u32)__builtin_popcountll(x);
#endif
  }
  // (the register-resident free mask serves batches of <= 64 slots; the two-wavefront large-network kernel never has that few and
  // keeps the mask out of its registers)
  LBFT_HD bool mask_slots() const { return QUAD ? true : LEAN2 ? false : P.scap <= 64; }
  LBFT_HD void snap_free_slot(u32 slot) {
    if (mask_slots()) snap_mask |= 1ULL << slot;
    else { st(OFFSFREE() + snap_free, slot); snap_free++; }
  }
  LBFT_HD i32 snap_alloc() {
    if (mask_slots()) {
      if (LBFT_UNLIKELY(snap_mask == 0)) { fault |= F_SNAP_OVERFLOW; return -1; }
      u32 slot = ctz64(snap_mask);
      snap_mask &= snap_mask - 1;
      u32 live = P.scap - popc64(snap_mask);
      if (live > maxsnap) maxsnap = live;
      return (i32)slot;
    }
    if (snap_free == 0) { fault |= F_SNAP_OVERFLOW; return -1; }
    snap_free--;
    u32 live = P.scap - snap_free;
    if (live > maxsnap) maxsnap = live;
    return (i32)ld(OFFSFREE() + snap_free);
  }
  // Reference count of a snapshot slot.  Small networks: a row of its own (OFFSREF).  Large networks (round 6): bits 16..31 of the slot's S_EPOCH word
  // (epochs stay below 2^16: bounded by the block capacity) -- the count then rides in the line every reader of the slot fetches anyway, instead of costing
  // a line of its own per notification / request / response (0.3-1.0 lines read and as many written per event: tests/tools/mem_lines.cpp).
  LBFT_HD bool refpack() const { return wide(); }
  LBFT_HD void snap_set_refs(u32 slot, u32 refs, u32 epoch) {  // `epoch`: what the slot's S_EPOCH word holds (the caller wrote it in this step)
    if (refpack()) st(sfw(slot, S_EPOCH), epoch | (refs << 16)); else st(OFFSREF() + slot, refs);
  }
  LBFT_HD void snap_release(u32 slot) {
    if (refpack()) { u32 w = ld(sfw(slot, S_EPOCH)); snap_release(slot, w >> 16, w & 0xffffu); }
    else snap_release(slot, ld(OFFSREF() + slot), 0);
  }
  LBFT_HD void snap_release(u32 slot, u32 refs, u32 epoch) {  // `refs` (and the slot's epoch) as loaded by the caller
    u32 r = refs - 1;
    snap_set_refs(slot, r, epoch);
    if (r == 0) snap_free_slot(slot);
  }

  // extension "lossy network": called right after a message's delay draw; true = the message is lost
  LBFT_HD bool lossy() const { return !C0 && !LEAN && (P.drop_ppm | P.part_size) != 0; }
  LBFT_HD bool net_lost(u32 a, u32 b) {
    if (!lossy()) return false;
    bool lost = false;
    if (P.drop_ppm) { u64 d = rng.next_u64(); lost = mulhi64(d, 1000000ULL) < (u64)P.drop_ppm; }
    if (P.part_size && clock >= P.part_start && clock < P.part_end && ((a < P.part_size) != (b < P.part_size))) lost = true;
    return lost;
  }
  LBFT_HD bool is_equivocator(u32 node) const { return !C0 && P.equiv != 0 && node % P.equiv == 0; }  // class 0: all honest
  // EpochConfiguration of the node's current epoch (extension "rotating voting rights": shifted by epoch * rot)
  // (32-bit arithmetic: epochs are bounded by the block capacity 65534 and rot < n <= 128; a 64-bit modulo is a ~200-instruction
  // software division inlined at every use)
  LBFT_HD u32 rights_shift(u32 node) const { return ROT() ? (nf(node, NF_EPOCH) * ROT()) % NN() : 0u; }
  LBFT_HD u32 weight(u32 node, u32 author) const {  // vector load from a small table
    if (UNITW()) return 1u;
    u32 i = author + rights_shift(node);
    return wtab[i >= NN() ? i - NN() : i];
  }

  // ---- leader / duration ----
  LBFT_HD u32 leader(u32 node, u32 round) const {
    u32 shift = rights_shift(node);
    // One load through a SELECTED table pointer, not `if (..) return leader_lds[round]; if (..) return P.leader_tab[round];`:
    // when the two pointer members happen to sit at the same offset of their structs, the optimiser merges the two
    // branches into one load through a phi of `this` and `&P`, after which neither struct is promoted to registers any
    // more (the whole simulator state silently moves to scratch memory; tests/test_abi.py guards the symptom).
    if (QUAD && LBFT_QUAD_LDS_ROUND_TABLES) {  // (no rotation: shift == 0; a 4-node run to clock 1000 stays far below the table's 1 024 rounds)
      if (LBFT_LIKELY(round < leader_lds_len)) return leader_lds[round];
      if (round < P.leader_len) return P.leader_tab[round];
      return wuni(compute_leader(P.weights, NN(), P.total_votes, round, shift));
    }
    if (LBFT_LIKELY(round < P.leader_len)) {
      const u8* tab = (round < leader_lds_len && shift == 0) ? leader_lds : P.leader_tab + (size_t)shift * P.leader_len;
      return wuni((u32)tab[round]);  // (a flat load counts as a source of divergence)
    }
    return wuni(compute_leader(P.weights, NN(), P.total_votes, round, shift));
  }

  // ---- SimulatedContext (simulated_context.rs:102-158): is the ledger state `blk` available? ----
  LBFT_HD bool state_available(u32 node, u32 blk) const {
    if (blk == nf(node, NF_LAST_COMMITTED_BLK)) return true;
    if (blk == 0) return false;
    Blk r = blk_get(blk);
    return state_pending(node, blk, r);
  }
  // pending_ledger_states.contains(state of blk).  B_PEND records that the node has EVER computed the state; a commit
  // removes the state from the map (simulated_context.rs:160-166), which is not recorded by clearing the bit (that
  // would cost the commit a read-modify-write of an old block's row) but recognised here: the k-th commit of a node is
  // the block of ledger depth k + 1 (every commit extends the previous one by one command), so a block has been
  // committed iff the node's log holds it at its depth.  Only blocks at or below the committed depth -- forks and
  // stragglers -- ever reach the log lookup.
  LBFT_HD bool state_pending(u32 node, u32 blk, const Blk& r) const {
    if (!bm_test(blk, r, B_PEND, node)) return false;
    u32 d = r.depth();
    if (LBFT_LIKELY(d > nf(node, NF_NCOMMITS))) return true;
    return ld(P.off_log + node * P.lcap + d - 1) != blk;
  }
  // RecordStoreState::compute_state (record_store.rs:426-454) + CommandExecutor::compute.
  // `rb` is the caller's copy of block b's record; its pending mask is updated in place.
  LBFT_HD bool compute_state(u32 node, u32 b, Blk& rb) const {
    u32 prev = rb.prev();
    u32 base = prev ? prev : nf(node, NF_INIT_STATE_BLK);
    if (!state_available(node, base)) return false;
    bm_set(b, rb, B_PEND, node);
    return true;
  }

  // ---- author sets of a node (NF_TC_MASK, NF_TO_MASK, NF_BAL0_AUTHORS, NF_BAL1_AUTHORS): authors 0..31 in
  // the cached fixed rows, authors >= 32 (n > 32 only) in extension rows behind the hcbr buffers ----
  LBFT_HD u32 am_idx(u32 f) const { return f == NF_TC_MASK ? 0u : f == NF_TO_MASK ? 1u : f == NF_BAL0_AUTHORS ? 2u : 3u; }
  LBFT_HD u32 amxw(u32 node, u32 f, u32 k) const { return nfw(node, NF_FIXED_WORDS + am_idx(f) * (MW() - 1) + k - 1); }
  // Words 1..3 of the four sets are staged in registers with the fixed rows (begin_node / end_node): a read-modify-write of
  // an extension row would otherwise be one dependent memory round trip per author >= 32 in every vote / timeout insertion
  // (64-node networks: half of all authors).  Indices are kept compile-time after unrolling (value selects, not dynamically
  // indexed stores) so that ax[][] lives in registers.
  mutable u32 ax[4][3];
  mutable u32 axdirty;  // bit (set * 3 + word - 1)
  LBFT_HD void ax_load(u32 node) const {
    axdirty = 0;
    if (!F_AX || !wide()) return;
    u32 base = nfw(node, NF_FIXED_WORDS);
    LBFT_UNROLL
    for (u32 i = 0; i < 4; i++) {
      LBFT_UNROLL
      for (u32 k = 0; k < 3; k++) {  // (unconditional loads of a clamped row, then a select: one burst with the fixed rows)
        u32 kk = k + 1 < MW() ? k : 0;
        u32 v = ld(base + i * (MW() - 1) + kk);
        ax[i][k] = k + 1 < MW() ? v : 0u;
      }
    }
  }
  LBFT_HD void ax_store(u32 node) const {
    if (!F_AX || !wide() || !axdirty) return;
    LBFT_UNROLL
    for (u32 i = 0; i < 4; i++) {
      LBFT_UNROLL
      for (u32 k = 0; k < 3; k++)
        if ((axdirty >> (i * 3 + k)) & 1u) st(nfw(node, NF_FIXED_WORDS + i * (MW() - 1) + k), ax[i][k]);
    }
  }
  LBFT_HD u32 ax_get(u32 i, u32 k) const {  // k = 1..3
    u32 v = 0;
    LBFT_UNROLL
    for (u32 ii = 0; ii < 4; ii++) {
      LBFT_UNROLL
      for (u32 kk = 0; kk < 3; kk++) v = (ii == i && kk + 1 == k) ? ax[ii][kk] : v;
    }
    return v;
  }
  LBFT_HD void ax_put(u32 i, u32 k, u32 v) const {
    LBFT_UNROLL
    for (u32 ii = 0; ii < 4; ii++) {
      LBFT_UNROLL
      for (u32 kk = 0; kk < 3; kk++) ax[ii][kk] = (ii == i && kk + 1 == k) ? v : ax[ii][kk];
    }
    axdirty |= 1u << (i * 3 + k - 1);
  }
  LBFT_HD u32 am_word(u32 node, u32 f, u32 k) const { return k == 0 ? nf(node, f) : F_AX ? ax_get(am_idx(f), k) : ld(amxw(node, f, k)); }
  LBFT_HD void am_set_word(u32 node, u32 f, u32 k, u32 v) const { if (k == 0) nfs(node, f, v); else if (F_AX) ax_put(am_idx(f), k, v); else st(amxw(node, f, k), v); }
  LBFT_HD bool am_test(u32 node, u32 f, u32 a) const {
    if (!wide() || a < 32) return (nf(node, f) >> a) & 1u;
    return (am_word(node, f, a >> 5) >> (a & 31u)) & 1u;
  }
  LBFT_HD void am_set(u32 node, u32 f, u32 a) const {
    if (!wide() || a < 32) nfs(node, f, nf(node, f) | (1u << a));
    else am_set_word(node, f, a >> 5, am_word(node, f, a >> 5) | (1u << (a & 31u)));
  }
  LBFT_HD void am_clear(u32 node, u32 f) const {
    nfs(node, f, 0);
    for (u32 k = 1; wide() && k < MW(); k++) am_set_word(node, f, k, 0);
  }
  LBFT_HD void am_copy(u32 node, u32 dst, u32 src) const {
    nfs(node, dst, nf(node, src));
    for (u32 k = 1; wide() && k < MW(); k++) am_set_word(node, dst, k, am_word(node, src, k));
  }

  // ---- RecordStoreState ----
  LBFT_HD void clear_ballot(u32 node) const {
    nfs(node, NF_BAL0_BLK, 0); nfs(node, NF_BAL0_WEIGHT, 0); am_clear(node, NF_BAL0_AUTHORS);
    nfs(node, NF_BAL1_BLK, 0); nfs(node, NF_BAL1_WEIGHT, 0); am_clear(node, NF_BAL1_AUTHORS);
  }
  LBFT_HD void update_current_round(u32 node, u32 round) const {  // record_store.rs:207-219
    if (round <= nf(node, NF_CUR_ROUND)) return;
    nfs(node, NF_CUR_ROUND, round);
    nfs(node, NF_PROPOSED_BLK, 0);
    am_clear(node, NF_TO_MASK);
    nfs(node, NF_TO_WEIGHT, 0);
    nfs(node, NF_ELECTION, 0);
    clear_ballot(node);
  }
  LBFT_HD void update_commit_3chain_round(u32 node, u32 b, const Blk& rb) const {  // record_store.rs:221-235
    if (!rb.prev() || !rb.pp()) return;
    u32 r3 = rb.round(), r2 = rb.prev_round(), r1 = rb.pp_round();
    if (r3 == r2 + 1 && r2 == r1 + 1 && r1 > nf(node, NF_HC_ROUND)) {
      nfs(node, NF_HC_ROUND, r1);
      nfs(node, NF_HCC_BLK, b);
    }
  }
  // verify_network_record + try_insert_network_record for a QC (record_store.rs:330-389,500-526).
  // Caller has already checked qc.epoch_id == node epoch (node.rs:151-167).
  LBFT_HD void insert_qc(u32 node, u32 b, Blk& rb) const {
    if (bm_test(b, rb, B_QC, node)) return;      // "QuorumCertificate was already inserted."
    if (!bm_test(b, rb, B_KNOWN, node)) return;  // "The certified block hash of a QC must be verified first."
    bm_set(b, rb, B_QC, node);                   // Q3: stored before the execution check
    if (!compute_state(node, b, rb)) return;  // bail!("I failed to execute a block with a QC ...")
    u32 r = rb.round();
    if (r > nf(node, NF_HQC_ROUND)) { nfs(node, NF_HQC_ROUND, r); nfs(node, NF_HQC_BLK, b); }
    update_current_round(node, r + 1);
    update_commit_3chain_round(node, b, rb);
  }
  LBFT_HD void insert_qc(u32 node, u32 b) const {
    Blk rb = blk_get(b);
    insert_qc(node, b, rb);
  }
  // Block (record_store.rs:263-291,466-476)
  LBFT_HD void insert_block(u32 node, u32 b, Blk& rb) const {
    if (bm_test(b, rb, B_KNOWN, node)) return;  // "Block was already inserted."
    u32 p = rb.prev();
    if (p) {  // "The previous QC (if any) must be verified first."
      Blk rp = blk_get(p);
      if (!bm_test(p, rp, B_QC, node)) return;
    }
    u32 r = rb.round();
    if (r == nf(node, NF_CUR_ROUND) && leader(node, r) == rb.author()) nfs(node, NF_PROPOSED_BLK, b);
    bm_set(b, rb, B_KNOWN, node);
  }
  LBFT_HD void insert_block(u32 node, u32 b) const {
    Blk rb = blk_get(b);
    insert_block(node, b, rb);
  }
  // Vote (record_store.rs:292-329,477-499).  Caller checked the epoch.
  LBFT_HD void insert_vote(u32 node, u32 author, u32 b, const Blk& rb) {
    if (!bm_test(b, rb, B_KNOWN, node)) return;
    if (rb.round() != nf(node, NF_CUR_ROUND)) return;
    if (am_test(node, NF_BAL0_AUTHORS, author) || am_test(node, NF_BAL1_AUTHORS, author)) return;  // one vote per author
    u32 b0 = nf(node, NF_BAL0_BLK), b1 = nf(node, NF_BAL1_BLK);
    bool ongoing = (nf(node, NF_ELECTION) & 0xff) == 0;
    // two ballot entries (block, weight, authors); field indices stay compile-time constants so that
    // the node cache remains in registers
    if (b0 == b || b0 == 0) {
      nfs(node, NF_BAL0_BLK, b);
      am_set(node, NF_BAL0_AUTHORS, author);
      if (ongoing) {
        u32 w = nf(node, NF_BAL0_WEIGHT) + weight(node, author);
        nfs(node, NF_BAL0_WEIGHT, w);
        if (w >= QUORUM()) nfs(node, NF_ELECTION, 1u | (b << 8));
      }
    } else if (b1 == b || b1 == 0) {
      nfs(node, NF_BAL1_BLK, b);
      am_set(node, NF_BAL1_AUTHORS, author);
      if (ongoing) {
        u32 w = nf(node, NF_BAL1_WEIGHT) + weight(node, author);
        nfs(node, NF_BAL1_WEIGHT, w);
        if (w >= QUORUM()) nfs(node, NF_ELECTION, 1u | (b << 8));
      }
    } else {
      fault |= F_BALLOT_OVERFLOW;  // (never with honest voters: two ballot entries hold any round's votes)
    }
  }
  // Timeout (record_store.rs:390-415,527-538).  Caller checked the epoch.
  LBFT_HD void insert_timeout(u32 node, u32 author, u32 round, u32 hcbr) const {
    if (hcbr > nf(node, NF_HQC_ROUND)) return;
    u32 cur = nf(node, NF_CUR_ROUND);
    if (round != cur) return;
    if (am_test(node, NF_TO_MASK, author)) return;
    am_set(node, NF_TO_MASK, author);
    u32 tc_sel = nf(node, NF_TC_SEL);
    hc_set(node, 1u - tc_sel, author, hcbr);
    u32 w = nf(node, NF_TO_WEIGHT) + weight(node, author);
    nfs(node, NF_TO_WEIGHT, w);
    if (w >= QUORUM()) {
      // the current timeouts become the timeout certificate (record_store.rs:532-534): swap the buffers
      // instead of copying n words; the other buffer's stale entries are masked by the now-empty TO mask
      am_copy(node, NF_TC_MASK, NF_TO_MASK);
      nfs(node, NF_TC_SEL, 1u - tc_sel);
      nfs(node, NF_HTC_ROUND, cur);
      update_current_round(node, cur + 1);
    }
  }
  // The timeouts of a notification (data_sync.rs:150-163), in author order; the hcbr words of up to four
  // authors are fetched in one burst before they are inserted.
  LBFT_HD void insert_timeouts(u32 node, u32 slot, u32 first_word, u32 mask, u32 round, u32 author0 = 0) const {
    insert_timeouts_at(node, sfw(slot, first_word), mask, round, author0);
  }
  // (`word0`: row of the first author's highest_certified_block_round -- a snapshot slot or an archived record store)
  LBFT_HD void insert_timeouts_at(u32 node, u32 word0, u32 mask, u32 round, u32 author0 = 0) const {
    // an author whose timeout the node already holds is rejected without side effects (record_store.rs:390-415) and inserting
    // one author never changes that for another: only the new ones are fetched
    mask &= ~am_word(node, NF_TO_MASK, author0 >> 5);
    constexpr u32 TB = BIG ? 8 : 4;  // hcbr words in flight per round trip
    while (mask) {
      u32 a[TB], h[TB], k = 0;
      LBFT_UNROLL
      for (u32 j = 0; j < TB; j++) {
        a[j] = 0; h[j] = 0;
        if (mask) { a[j] = author0 + ctz32(mask); mask &= mask - 1; h[j] = ld(word0 + a[j]); k = j + 1; }
      }
      LBFT_UNROLL
      for (u32 j = 0; j < TB; j++)
        if (j < k) insert_timeout(node, a[j], round, h[j]);
    }
  }
  // RecordStore::proposed_block (record_store.rs:611-634): block id or 0
  LBFT_HD u32 proposed_block(u32 node) const {
    if (nf(node, NF_EPOCH) != nf(node, NF_PM_EPOCH) || nf(node, NF_CUR_ROUND) != nf(node, NF_PM_ROUND)) return 0;
    if (nf(node, NF_PM_LEADER) == LBFT_NO_LEADER) return 0;
    return nf(node, NF_PROPOSED_BLK);
  }
  // propose_block (record_store.rs:655-674) + CommandFetcher::fetch (simulated_context.rs:116-125)
  LBFT_HD void propose_block(u32 node, u32 prev_blk, i64 local_clock) {
    u32 cmd = nf(node, NF_NEXT_CMD);
    nfs(node, NF_NEXT_CMD, cmd + 1);
    if (LBFT_UNLIKELY(nblocks >= P.bcap || nblocks >= 0xfffeu)) { fault |= F_BLOCK_OVERFLOW; return; }
    u32 b = ++nblocks;
    u32 base = prev_blk ? prev_blk : nf(node, NF_INIT_STATE_BLK);
    Blk rb;
    rb.xk = 0; rb.x[0] = rb.x[1] = rb.x[2] = 0;
    rb.w[B_ROUND] = nf(node, NF_CUR_ROUND);
    rb.w[B_LINK] = prev_blk | (node << 16);
    rb.w[B_PREV_ROUND] = 0; rb.w[B_PP] = 0; rb.w[B_PP_ROUND] = 0; rb.w[B_DEPTH] = 1;
    if (base) {  // denormalised ancestry: previous_round / second_previous_round (record_store.rs:588-609)
      Blk rp = blk_get(base);
      rb.w[B_DEPTH] = rp.depth() + 1;
      if (prev_blk) { rb.w[B_PREV_ROUND] = rp.round(); rb.w[B_PP] = rp.prev() | (rp.pp() << 16); rb.w[B_PP_ROUND] = rp.prev_round(); }
    }
    rb.w[B_EPOCH] = nf(node, NF_EPOCH);
    rb.w[B_KNOWN] = 0; rb.w[B_QC] = 0; rb.w[B_PEND] = 0;
    LBFT_UNROLL
    for (u32 f = 0; f < BC_WORDS; f++) bfs(b, f, rb.w[f]);
    bfs(b, B_TIME, (u32)(i32)local_clock);
    bfs(b, B_CMD, cmd);
    bfs(b, B_VOTERS, 0);
    for (u32 k = 0; wide() && k < 4 * (MW() - 1); k++) bfs(b, B_WORDS + k, 0);
    blk_cache_insert(b, rb);
    blw_fill(b, rb);
    insert_block(node, b, rb);
  }
  // create_vote (record_store.rs:676-700)
  LBFT_HD bool create_vote(u32 node, u32 b, Blk& rb) {
    if (!compute_state(node, b, rb)) return false;
    insert_vote(node, node, b, rb);
    return true;
  }
  // check_for_new_quorum_certificate (record_store.rs:702-738)
  LBFT_HD bool check_for_new_qc(u32 node) const {
    u32 e = nf(node, NF_ELECTION);
    if ((e & 0xff) != 1) return false;
    u32 b = e >> 8;
    Blk rb = blk_get(b);
    if (rb.author() != node) return false;
    nfs(node, NF_ELECTION, 2);
    // the QC's votes = the current votes for the winning (block, state) (record_store.rs:716-727): the ballot entry of b
    bool first = nf(node, NF_BAL0_BLK) == b;  // (field indices stay compile-time constants: the node cache lives in registers)
    for (u32 k = 0; k < MW(); k++) {
      u32 v = first ? am_word(node, NF_BAL0_AUTHORS, k) : am_word(node, NF_BAL1_AUTHORS, k);
      bfs(b, k == 0 ? (u32)B_VOTERS : B_WORDS + 3 * (MW() - 1) + k - 1, v);
    }
    insert_qc(node, b, rb);
    return true;
  }

  // ---- Pacemaker (pacemaker.rs:111-124,142-207) ----
  struct PmActions { bool propose; u32 propose_prev; bool create_timeout; u32 timeout_round; i32 send_to; bool broadcast; bool query_all; i64 next; };

  LBFT_HD i64 duration(u32 node, u32 round) {
    u32 hc = nf(node, NF_HC_ROUND);
    u32 hccr = hc > 0 ? hc + 2 : 0;
    if (LBFT_UNLIKELY(round <= hccr)) { fault |= F_INTERNAL; return 0; }
    u32 k = round - hccr;
    if (LBFT_UNLIKELY(k >= P.dur_len)) { fault |= F_DURATION_TABLE; k = P.dur_len - 1; }
    if (QUAD && LBFT_QUAD_LDS_ROUND_TABLES) {
      if (LBFT_LIKELY(k < dur_lds_len)) return dur_lds[k];
      return P.dur_tab[k];
    }
    const i64* tab = k < dur_lds_len ? dur_lds : P.dur_tab;  // (a selected pointer: see leader())
    return (i64)wuni((u64)tab[k]);
  }
  LBFT_HD PmActions update_pacemaker(u32 node, i64 lqat, i64 lclock) {
    PmActions a;
    a.propose = false; a.propose_prev = 0; a.create_timeout = false; a.timeout_round = 0;
    a.send_to = -1; a.broadcast = false; a.query_all = false; a.next = LBFT_NEVER;
    u32 hqc = nf(node, NF_HQC_ROUND), htc = nf(node, NF_HTC_ROUND);
    u32 ar = (hqc > htc ? hqc : htc) + 1;
    u32 epoch = nf(node, NF_EPOCH), pe = nf(node, NF_PM_EPOCH);
    u32 pm_round = nf(node, NF_PM_ROUND);
    u32 pm_leader = nf(node, NF_PM_LEADER);
    i64 start = (i64)(i32)nf(node, NF_PM_START);
    i64 dur = (i64)(nf(node, NF_PM_DUR_LO) | ((u64)nf(node, NF_PM_DUR_HI) << 32));
    if (epoch > pe || (epoch == pe && ar > pm_round)) {
      pm_round = ar;
      pm_leader = leader(node, ar);
      start = lclock;
      dur = duration(node, ar);
      nfs(node, NF_PM_EPOCH, epoch); nfs(node, NF_PM_ROUND, ar); nfs(node, NF_PM_LEADER, pm_leader);
      nfs(node, NF_PM_START, (u32)(i32)lclock);
      nfs(node, NF_PM_DUR_LO, (u32)(u64)dur); nfs(node, NF_PM_DUR_HI, (u32)((u64)dur >> 32));
      if (pm_leader != node) a.send_to = (i32)pm_leader;
    }
    if (pm_leader == node && proposed_block(node) == 0) {
      a.propose = true;
      a.propose_prev = nf(node, NF_HQC_BLK);
      a.broadcast = true;
      a.next = lclock;
    }
    // has_timeout(local_author, active_round) (record_store.rs:651-653); `ar` is the recomputed
    // active round exactly as in the reference (pacemaker.rs:183)
    bool has_timeout = (ar == nf(node, NF_CUR_ROUND)) && am_test(node, NF_TO_MASK, node);
    if (!has_timeout) {
      i64 deadline = (i64)((u64)start + (u64)dur);
      if (lclock >= deadline) { a.create_timeout = true; a.timeout_round = ar; a.broadcast = true; }
      else if (deadline < a.next) a.next = deadline;
    } else {
      i64 period = f64_to_i64_sat(P.lambda * (double)dur);
      i64 qd = (i64)((u64)lqat + (u64)period);
      if (lclock >= qd) { a.query_all = true; qd = (i64)((u64)lclock + (u64)period); }
      if (qd < a.next) a.next = qd;
    }
    return a;
  }

  // ---- CommitTracker::update_tracker (node.rs:364-396) ----
  LBFT_HD void update_tracker(u32 node, i64 lqat, i64 lclock, bool& query_all, i64& next) const {
    u32 epoch = nf(node, NF_EPOCH);
    i64 lct = (i64)(i32)nf(node, NF_TR_LCT);
    if (epoch > nf(node, NF_TR_EPOCH)) {
      nfs(node, NF_TR_EPOCH, epoch);
      nfs(node, NF_TR_HCR, nf(node, NF_HC_ROUND));
      lct = lclock;
      nfs(node, NF_TR_LCT, (u32)(i32)lclock);
    } else {
      u32 hcr = nf(node, NF_HC_ROUND);
      if (hcr > nf(node, NF_TR_HCR)) {
        nfs(node, NF_TR_HCR, hcr);
        lct = lclock;
        nfs(node, NF_TR_LCT, (u32)(i32)lclock);
      }
    }
    i64 deadline = (i64)((u64)(lct > lqat ? lct : lqat) + (u64)P.tci);
    query_all = false;
    if (lclock >= deadline) { query_all = true; deadline = (i64)((u64)lclock + (u64)P.tci); }
    next = deadline;
  }

  // ---- NodeState::process_commits (node.rs:313-350) + StateFinalizer::commit ----
  LBFT_HD void process_commits(u32 node) {
    u32 after = nf(node, NF_TR_HCR);
    // The walk below starts at the grandparent of the commit-certificate block, whose round is exactly
    // highest_committed_round (update_commit_3chain_round sets both together): nothing new to commit.
    if (nf(node, NF_HC_ROUND) <= after) return;
    LBFT_STAT(11);
    // committed_states_after (record_store.rs:557-574): from the grandparent of the commit-certificate
    // block back to the first block whose round is <= `after`
    Blk rh = blk_get(nf(node, NF_HCC_BLK));
    u32 start = rh.pp(), k = 1;
    // Its round is highest_committed_round > after: at least this block commits.  The usual case is that it extends the
    // node's last commit directly -- then everything the commit needs is in the commit-certificate block's record
    // (grandparent, great-grandparent, depth) and no old block row is touched: its parent IS the last committed block
    // (so its round is `after` and the walk below would stop at once), and its state is pending (the node computed it
    // before it could execute the parent, record_store.rs:426-454, and it has not been committed yet).
    {
      u32 base0 = rh.ppp() ? rh.ppp() : nf(node, NF_INIT_STATE_BLK);
      if (base0 == nf(node, NF_LAST_COMMITTED_BLK)) { LBFT_STAT(12); commit_block(node, start, rh.depth() - 2); return; }
    }
    Blk r0;
    LBFT_UNROLL
    for (u32 f = 0; f < BC_WORDS; f++) r0.w[f] = 0;
    r0.xk = 0; r0.x[0] = r0.x[1] = r0.x[2] = 0;
    if (LBFT_COMMIT_CHAIN && !wide()) {
      constexpr u32 CAP = 4;
      u32 cid[CAP], clink[CAP], cdep[CAP], cpend[CAP];
      u32 kk = 0, x = start;
      bool more = true;
      LBFT_UNROLL
      for (u32 i = 0; i < CAP; i++) {
        cid[i] = 0; clink[i] = 0; cdep[i] = 0; cpend[i] = 0;
        if (more) {
          const u32 sb = boff(bfw(x, 0));
          const u32 link = ldf(sb, B_LINK), pr = ldf(sb, B_PREV_ROUND), d = ldf(sb, B_DEPTH), pe = ldf(sb, B_PEND);
          cid[i] = x; clink[i] = link; cdep[i] = d; cpend[i] = pe; kk = i + 1;
          x = link & 0xffffu;
          more = x != 0 && pr > after;  // (the parent commits too while its round -- this block's previous_round -- is above `after`)
        }
      }
      if (LBFT_LIKELY(!more)) {
        for (u32 j = kk; j-- > 0;) {  // oldest first
          const u32 y = j == 0 ? cid[0] : j == 1 ? cid[1] : j == 2 ? cid[2] : cid[3];
          Blk ry = r0;
          ry.w[B_LINK] = j == 0 ? clink[0] : j == 1 ? clink[1] : j == 2 ? clink[2] : clink[3];
          ry.w[B_DEPTH] = j == 0 ? cdep[0] : j == 1 ? cdep[1] : j == 2 ? cdep[2] : cdep[3];
          ry.w[B_PEND] = j == 0 ? cpend[0] : j == 1 ? cpend[1] : j == 2 ? cpend[2] : cpend[3];
          if (LBFT_UNLIKELY(!state_pending(node, y, ry))) { fault |= F_COMMIT_UNKNOWN_STATE; return; }
          const u32 prev = ry.prev();
          const u32 base = prev ? prev : nf(node, NF_INIT_STATE_BLK);
          if (LBFT_UNLIKELY(base != nf(node, NF_LAST_COMMITTED_BLK))) { fault |= F_COMMIT_NOT_SUCCESSOR; return; }
          if (commit_block(node, y, ry.depth())) break;
        }
        return;
      }
      // (a chain longer than CAP: the general walk below)
    }
    {  // an old block that nothing else will look at again: straight from its row, not through the cache
      u32 sb = boff(bfw(start, 0));
      r0.w[B_LINK] = ldf(sb, B_LINK); r0.w[B_PREV_ROUND] = ldf(sb, B_PREV_ROUND);
      r0.w[B_DEPTH] = ldf(sb, B_DEPTH); r0.w[B_PEND] = ldf(sb, B_PEND);
    }
    // (the walks below visit OLD blocks that nothing else will look at again: straight from their rows -- the rows are always current,
    // every cache update is written through -- instead of through the register cache, where each of them evicted a hot record:
    // 47 of a 4-node network's 119 cache misses were these lookups, round 4)
    for (u32 x = (r0.prev() && r0.prev_round() > after) ? r0.prev() : 0; x;) {
      u32 xr, xp;
      chain_fetch(x, xr, xp);
      if (xr <= after) break;
      k++;
      x = xp;
    }
    for (u32 j = k; j-- > 0;) {  // oldest first
      u32 y = start;
      for (u32 s = 0; s < j; s++) { u32 yr, yp; chain_fetch(y, yr, yp); y = yp; }
      Blk ry = r0;
      if (j != 0) {
        u32 sb = boff(bfw(y, 0));
        ry.w[B_LINK] = ldf(sb, B_LINK); ry.w[B_PREV_ROUND] = ldf(sb, B_PREV_ROUND);
        ry.w[B_DEPTH] = ldf(sb, B_DEPTH); ry.w[B_PEND] = ldf(sb, B_PEND);
      }
      // SimulatedContext::commit (simulated_context.rs:160-185)
      if (LBFT_UNLIKELY(!state_pending(node, y, ry))) { fault |= F_COMMIT_UNKNOWN_STATE; return; }
      u32 prev = ry.prev();
      u32 base = prev ? prev : nf(node, NF_INIT_STATE_BLK);
      if (LBFT_UNLIKELY(base != nf(node, NF_LAST_COMMITTED_BLK))) { fault |= F_COMMIT_NOT_SUCCESSOR; return; }
      if (commit_block(node, y, ry.depth())) break;
    }
  }
  // past_record_stores.insert(self.epoch_id, old_record_store) (node.rs:338-340), kept IN FULL when the batch asked for it
  // (lbft_batch_keep_retired_stores: what save_node needs for a node that has changed epoch): the node's rows -- fixed words, hcbr
  // buffers, set extension words -- copied verbatim into the archive entry of the epoch being left.  The blocks and certificates of
  // the retired store need no copy: they are the block rows of that epoch whose KNOWN / QC bit the node holds (never set again later).
  LBFT_HD void retire_store(u32 node) {
#if defined(LBFT_NO_RETIRE)
    return;
#endif
    if (QUAD || !P.rarch_words) return;  // (the compile-time-specialised kernel leaves batches that archive to the generic one: sim_quad)
    u32 old_epoch = nf(node, NF_EPOCH);
    if (old_epoch >= P.ecap) { fault |= F_EPOCH_OVERFLOW; return; }
    u32 base = P.off_rarch + (node * P.ecap + old_epoch) * P.rarch_words;
    // (the fixed words come from the register copy, everything behind them from the node's rows: the hcbr buffers -- LDS-resident for
    // networks of <= 4 nodes in class 0 -- and the set extension words, which are flushed first)
    LBFT_UNROLL
    for (u32 f = 0; f < NF_FIXED_WORDS; f++) st(base + f, cwg(f));
    ax_store(node);
    axdirty = 0;
    const u32 row = nfw(node, NF_FIXED_WORDS), tail = P.rarch_words - NF_FIXED_WORDS;
    if (hc_lds()) {
      for (u32 buf = 0; buf < 2; buf++)
        for (u32 a = 0; a < NN(); a++) st(base + NF_FIXED_WORDS + buf * NN() + a, hc_get(node, buf, a));
    } else {
      for (u32 k = 0; k < tail; k++) st(base + NF_FIXED_WORDS + k, ld(row + k));
    }
  }
  // The tail of SimulatedContext::commit + the epoch switch of process_commits (node.rs:331-348) for one block whose
  // state is pending and extends the last commit.  Returns true when no further block may be committed in this call
  // (epoch change or full log).
  LBFT_HD bool commit_block(u32 node, u32 y, u32 depth) {
    {
      nfs(node, NF_LAST_COMMITTED_BLK, y);
      u32 nc = nf(node, NF_NCOMMITS);
      if (LBFT_UNLIKELY(nc >= P.lcap)) { fault |= F_LOG_OVERFLOW; return true; }
      st(P.off_log + node * P.lcap + nc, y);
      nfs(node, NF_NCOMMITS, nc + 1);
      // read_epoch_id (simulated_context.rs:199-207)
      // epoch = depth / commands_per_epoch; the (software) 64-bit division only runs when a boundary is crossed
      if (LBFT_UNLIKELY((u64)depth >= ((u64)nf(node, NF_EPOCH) + 1) * P.cpe)) {
        // the epoch switch itself (node.rs:331-348) runs once process_commits has returned (update_node: epoch_switch) -- ONE site, and
        // one with few live registers, instead of a copy inside each commit path
        sw_epoch = (u32)wuni(udiv64((u64)depth, P.cpe)) + 1u;
        sw_blk = y;
        return true;
      }
    }
    return false;
  }

  // fresh RecordStoreState for the new epoch (node.rs:331-348, record_store.rs:169-198); the retired one is archived first
  u32 sw_epoch, sw_blk;  // pending epoch switch of the current event's node: new epoch + 1 (0: none), the block whose commit ended the old one
  LBFT_HD void epoch_switch(u32 node) {
    const u32 new_epoch = sw_epoch - 1u, y = sw_blk;
    sw_epoch = 0;
    if (q1()) {  // the store being retired stays readable for peers that ask later (past_record_stores, node.rs:43,339)
      u32 old_epoch = nf(node, NF_EPOCH);
      if (old_epoch < P.ecap) write_store_snapshot(node, arch_base(node, old_epoch)); else fault |= F_EPOCH_OVERFLOW;
    }
    retire_store(node);
    nfs(node, NF_PREV_EPOCH_HCC, nf(node, NF_HCC_BLK));  // ... and its commit certificate is what notifications forward (Q2 fixed)
    nfs(node, NF_EPOCH, new_epoch);
    nfs(node, NF_INIT_STATE_BLK, y);
    nfs(node, NF_PROPOSED_BLK, 0);
    nfs(node, NF_HQC_ROUND, 0); nfs(node, NF_HQC_BLK, 0); nfs(node, NF_HTC_ROUND, 0);
    nfs(node, NF_CUR_ROUND, 1); nfs(node, NF_HC_ROUND, 0); nfs(node, NF_HCC_BLK, 0);
    am_clear(node, NF_TC_MASK); am_clear(node, NF_TO_MASK); nfs(node, NF_TO_WEIGHT, 0);
    nfs(node, NF_ELECTION, 0);
    clear_ballot(node);
    nfs(node, NF_LVR, 0); nfs(node, NF_LOCKED, 0);
  }

  // ---- NodeState::update_node (node.rs:240-304) ----
  // `tail_only`: only process_commits + update_tracker, results discarded -- the step between two epochs of a response
  // (data_sync.rs:226-236, see handle_response_epoch)
  LBFT_HD Actions update_node(u32 node, i64 lclock, bool tail_only = false) {
    i64 lqat = (i64)(i32)nf(node, NF_LQAT);
    Actions act;
    act.next = LBFT_NEVER; act.send_to = -1; act.broadcast = false; act.query_all = false;
    if (!tail_only) {
    PmActions pa = update_pacemaker(node, lqat, lclock);
    LBFT_UMARK(6);
    act.next = pa.next; act.send_to = pa.send_to; act.broadcast = pa.broadcast; act.query_all = pa.query_all;
    // process_pacemaker_actions (node.rs:179-202)
    LBFT_STAT(5);
    if (pa.create_timeout) {
      LBFT_STAT(6);
      insert_timeout(node, node, pa.timeout_round, nf(node, NF_HQC_ROUND));  // create_timeout record_store.rs:636-649
      u32 lvr = nf(node, NF_LVR);
      if (pa.timeout_round > lvr) nfs(node, NF_LVR, pa.timeout_round);
    }
    if (pa.propose) {
      LBFT_STAT(7);
      propose_block(node, pa.propose_prev, lclock);
      // extension (E1): an equivocator proposes a second block B on the same previous QC; B = A + 1 becomes its
      // current proposed block, so the twin of an equivocator's own proposal is always "proposed block - 1"
      if (is_equivocator(node)) propose_block(node, pa.propose_prev, lclock);
    }
    LBFT_UMARK(7);
    // vote
    u32 pb = proposed_block(node);
    if (pb) {
      LBFT_STAT(8);
      Blk rpb = blk_get(pb);
      u32 br = rpb.round();
      u32 prev_round = rpb.prev_round();  // previous_round (record_store.rs:588-598)
      u32 locked = nf(node, NF_LOCKED);
      if (br > nf(node, NF_LVR) && prev_round >= locked) {
        nfs(node, NF_LVR, br);
        u32 second_prev = rpb.pp_round();  // second_previous_round (record_store.rs:600-609)
        if (second_prev > locked) nfs(node, NF_LOCKED, second_prev);
        LBFT_STAT(9);
        if (create_vote(node, pb, rpb)) act.send_to = (i32)rpb.author();
      }
    }
    LBFT_UMARK(8);
    if (check_for_new_qc(node)) { LBFT_STAT(10); act.broadcast = true; act.next = lclock; }
    LBFT_UMARK(9);
    }
    process_commits(node);
    if (LBFT_UNLIKELY(sw_epoch != 0)) epoch_switch(node);
    LBFT_MARK(27);
    bool tq; i64 tnext;
    update_tracker(node, lqat, lclock, tq, tnext);
    if (!tail_only) {
      act.query_all = act.query_all || tq;
      if (tnext < act.next) act.next = tnext;
      if (act.query_all) nfs(node, NF_LQAT, (u32)(i32)lclock);
    }
    LBFT_UMARK(10);
    return act;
  }

  // ---- (round 6) would update_node(node, lclock) change anything?  Evaluated on the node's staged fixed words alone (no block record): true = PROVABLY
  // nothing -- the pacemaker's epoch and round stand (pacemaker.rs:160-178), there is nothing to propose (:180-187), neither the timeout deadline nor the
  // query-all deadline has passed (:188-205), no vote is due (the proposed block's round IS the current round, record_store.rs proposed_block, and it is not
  // above latest_voted_round: node.rs:259-262 fails whatever the locked round says), no election was won (record_store.rs:702-738), nothing is left to commit
  // (node.rs:313-350: highest_committed_round is the tracker's), and the tracker neither moves nor asks for a query-all (node.rs:364-396).  `next` = the
  // next_scheduled_update the call returns then.  false = anything else: the caller takes the ordinary path (conservative wherever a block record would
  // have to be read to decide).  The response runs (coop_responses) rest on it: an event whose update is a no-op touches the node's timer words only.
  LBFT_HD bool update_is_noop(u32 node, i64 lclock, i64& next) const {
    const i64 lqat = (i64)(i32)nf(node, NF_LQAT);
    const u32 hqc = nf(node, NF_HQC_ROUND), htc = nf(node, NF_HTC_ROUND);
    const u32 ar = (hqc > htc ? hqc : htc) + 1;
    const u32 epoch = nf(node, NF_EPOCH);
    bool ok = epoch == nf(node, NF_PM_EPOCH) && ar == nf(node, NF_PM_ROUND);
    const u32 pb = proposed_block(node);
    ok = ok && !(nf(node, NF_PM_LEADER) == node && pb == 0);
    const i64 start = (i64)(i32)nf(node, NF_PM_START);
    const i64 dur = (i64)(nf(node, NF_PM_DUR_LO) | ((u64)nf(node, NF_PM_DUR_HI) << 32));
    const bool has_timeout = (ar == nf(node, NF_CUR_ROUND)) && am_test(node, NF_TO_MASK, node);
    i64 nx;
    if (!has_timeout) nx = (i64)((u64)start + (u64)dur);
    else nx = (i64)((u64)lqat + (u64)f64_to_i64_sat(P.lambda * (double)dur));
    ok = ok && lclock < nx;
    ok = ok && !(pb != 0 && nf(node, NF_CUR_ROUND) > nf(node, NF_LVR));
    ok = ok && (nf(node, NF_ELECTION) & 0xffu) != 1u;
    ok = ok && epoch <= nf(node, NF_TR_EPOCH) && nf(node, NF_HC_ROUND) <= nf(node, NF_TR_HCR);
    const i64 lct = (i64)(i32)nf(node, NF_TR_LCT);
    const i64 dl = (i64)((u64)(lct > lqat ? lct : lqat) + (u64)P.tci);
    ok = ok && lclock < dl;
    next = dl < nx ? dl : nx;
    return ok;
  }

  // ---- DataSyncNode::create_notification (data_sync.rs:82-111) into snapshot slot ----
  // hcbr words of the authors in `mask` (author = author0 + bit): node buffer -> snapshot, four loads in flight at a
  // time (a load-store-load-store chain would be one memory round trip per author)
  LBFT_HD void copy_hcbr(u32 node, u32 slot, u32 mask, u32 author0, u32 buf, u32 snap_word0) const { copy_hcbr_to(node, sfw(slot, snap_word0), mask, author0, buf); }
  LBFT_HD void copy_hcbr_to(u32 node, u32 dst_word0, u32 mask, u32 author0, u32 buf) const {
    constexpr u32 B = BIG ? LBFT_HCBR_BATCH : 4;  // loads in flight per round trip (large networks copy dozens of words per notification)
    while (mask) {
      u32 a[B], h[B], k = 0;
      LBFT_UNROLL
      for (u32 j = 0; j < B; j++) {
        a[j] = 0; h[j] = 0;
        if (mask) { a[j] = author0 + ctz32(mask); mask &= mask - 1; h[j] = hc_get(node, buf, a[j]); k = j + 1; }
      }
      LBFT_UNROLL
      for (u32 j = 0; j < B; j++)
        if (j < k) st(dst_word0 + a[j], h[j]);
    }
  }
  // ---- quirks bit 0 (reference quirk Q1 fixed): requests are answered by the PEER (bft-driver/src/core.rs:174-178
  // instead of simulator.rs:446) with the records the requester lacks (data_sync.rs:183-207, record_store.rs:766-831).
  // All records are immutable and live in the instance's block pool, so a response is described by the certificates
  // heading the peer's chains plus its timeouts and proposed block -- the same words as a notification snapshot. ----
  LBFT_HD u32 sqw(u32 base, u32 k) const { return base + S_FIXED_WORDS + 2 * NN() + 2 * (MW() - 1) + k; }
  // RecordStoreState as seen by unknown_records: written into `base` (a snapshot slot or an epoch-archive entry)
  // `skip_hcbr`: the caller has the timeouts' hcbr words copied by all lanes of the wavefront (coop_copy_hcbr_to)
  LBFT_HD void write_store_snapshot(u32 node, u32 base, bool skip_hcbr = false) const {
    st(base + S_EPOCH, nf(node, NF_EPOCH));
    st(base + S_CERTS, nf(node, NF_HCC_BLK) | (nf(node, NF_HQC_BLK) << 16));
    st(base + S_PROP_VOTE, nf(node, NF_PROPOSED_BLK));  // current_proposed_block, whoever proposed it
    u32 htc = nf(node, NF_HTC_ROUND);
    u32 xt = 0, xo = 0;
    for (u32 k = 1; wide() && k < MW(); k++) { xt |= htc ? am_word(node, NF_TC_MASK, k) : 0u; xo |= am_word(node, NF_TO_MASK, k); }
    st(base + S_TC_ROUND, htc | (xt ? LBFT_S_XFLAG : 0u));
    st(base + S_TO_ROUND, nf(node, NF_CUR_ROUND) | (xo ? LBFT_S_XFLAG : 0u));
    u32 tc_sel = nf(node, NF_TC_SEL);
    for (u32 k = 0; k < MW(); k++) {
      u32 tk = htc ? am_word(node, NF_TC_MASK, k) : 0, ok = am_word(node, NF_TO_MASK, k);
      st(k == 0 ? base + S_TC_MASK : base + S_FIXED_WORDS + 2 * NN() + (k - 1), tk);
      st(k == 0 ? base + S_TO_MASK : base + S_FIXED_WORDS + 2 * NN() + (MW() - 1) + (k - 1), ok);
      // (several loads in flight per round trip: a load-store-load-store chain is one memory round trip per author)
      if (!skip_hcbr) {
        copy_hcbr_to(node, base + S_FIXED_WORDS, tk, 32 * k, tc_sel);
        copy_hcbr_to(node, base + S_FIXED_WORDS + NN(), ok, 32 * k, 1u - tc_sel);
      }
    }
  }
  LBFT_HD u32 arch_base(u32 node, u32 epoch) const { return P.off_arch + (node * P.ecap + epoch) * SWORDS(); }
  // known_quorum_certificate_rounds (record_store.rs:766-799) of the requester: the rounds at positions 0, 1, 3, 7, 15, ... of
  // the chains below its highest quorum certificate and its highest commit certificate.  unknown_records asks "is round R
  // known?" for the rounds of the peer's chains, which only ever DEScend along a chain -- so each (peer chain, requester
  // chain) pair keeps a cursor that moves down the requester's chain monotonically (one light two-word fetch per block,
  // not through the block cache: nothing else looks at these old records) instead of restarting from the top for every
  // query: O(gap) block fetches per response instead of O(gap^2).
  struct Snap { u32 w[S_FIXED_WORDS]; u32 refs; u32 to_hcbr[4]; };  // (see load_snapshot)
  struct Resp { u32 epoch, certs, prop, req_epoch, req_certs; u32 tc_round, to_round, tc_mask0, to_mask0; };  // (see load_response)
  struct KnownCursor { u32 x, i, xr; };  // requester-chain block x at position i with round xr (x == 0: end of the chain)
  LBFT_HD void chain_fetch(u32 x, u32& round, u32& prev) const {
    u32 bb = boff(bfw(x, 0));
    round = ldf(bb, B_ROUND); prev = ldf(bb, B_LINK) & 0xffffu;
  }
  LBFT_HD KnownCursor known_start(u32 x) const {
    KnownCursor c; c.x = x; c.i = 0; c.xr = 0;
    if (x) { u32 pv; chain_fetch(x, c.xr, pv); }
    return c;
  }
  LBFT_HD bool known_at(KnownCursor& c, u32 round) const {
    while (c.x && c.xr > round) {
      u32 r, pv;
      chain_fetch(c.x, r, pv);
      c.x = pv; c.i++;
      c.xr = 0;
      if (pv) { u32 pv2; chain_fetch(pv, c.xr, pv2); }
    }
    return c.x != 0 && c.xr == round && ((c.i & (c.i + 1)) == 0);
  }
  // unknown_records (record_store.rs:801-831) of the store described at `base`, inserted into node's current store in
  // the order the reference sends them: (block, QC) pairs by ascending round, the timeouts, the proposed block.
  // `rp` / `have`: the words of `base` that decide what is walked, already fetched (load_response: the response's own store; an
  // archived store of an earlier epoch is read from its rows -- and so are the timeout words, which are needed last).  Most responses carry nothing the requester lacks (16 384 x 64 live: 325 new
  // records in 24.6 k responses per network), so the path to "nothing to insert" is two round trips: everything that decides it --
  // the heads of the peer's two chains (round + previous block: two words each, not through the block cache), the heads of the
  // requester's two chains, the "known" word of the proposed block -- is fetched in ONE burst.
  LBFT_HD void insert_unknown_records(u32 node, u32 base, bool filter, u32 k_hqc, u32 k_hcc, const Resp& rp, bool have) {
    u32 certs = have ? rp.certs : ld(base + S_CERTS);
    u32 x1 = certs >> 16, x2 = certs & 0xffffu, cnt = 0;
    const u32 pb = (have ? rp.prop : ld(base + S_PROP_VOTE)) & 0xffffu;
    u32 r1r = 0, r1p = 0, r2r = 0, r2p = 0, pbk = 0;
    if (LBFT_RESP_FAST && filter) {
      // known_at() answers true at position 0 of the requester's chain for exactly these: the walk of that peer chain ends before it starts
      if (x1 && (x1 == k_hqc || x1 == k_hcc)) x1 = 0;
      if (x2 && (x2 == k_hqc || x2 == k_hcc)) x2 = 0;
    }
    const bool walk = !LBFT_RESP_FAST || x1 != 0 || x2 != 0;
    if (x1) chain_fetch(x1, r1r, r1p);
    if (x2) chain_fetch(x2, r2r, r2p);
    if (pb) {
      bool got = false;
      if (LBFT_RESP_FAST && BLW && bl && (!wide() || node < 32)) {
        const u32 e = pb & (bl_n - 1u);
        if (bl[blx(e)] == pb) { pbk = bl[blx(bl_n + e * BC_WORDS + B_KNOWN)]; got = true; }
      }
      if (!got) pbk = ld((!wide() || node < 32) ? bfw(pb, B_KNOWN) : bxw(pb, B_KNOWN, node >> 5));
    }
    // cursors [peer chain][requester chain]
    KnownCursor q1q = known_start((filter && walk) ? k_hqc : 0), q1c = known_start((filter && walk) ? k_hcc : 0), q2q = q1q, q2c = q1c;
    bool fresh1 = false, fresh2 = false;  // x1 / x2 moved to a block that has not been looked at yet (the heads have)
    if (x1 && filter && (known_at(q1q, r1r) || known_at(q1c, r1r))) x1 = 0;
    if (x2 && filter && (known_at(q2q, r2r) || known_at(q2c, r2r))) x2 = 0;
    for (;;) {  // util.rs merge_sort of the two chains by descending round, identical certificates once
      if (x1 && fresh1) { chain_fetch(x1, r1r, r1p); fresh1 = false; if (filter && (known_at(q1q, r1r) || known_at(q1c, r1r))) x1 = 0; }
      if (x2 && fresh2) { chain_fetch(x2, r2r, r2p); fresh2 = false; if (filter && (known_at(q2q, r2r) || known_at(q2c, r2r))) x2 = 0; }
      if (!x1 && !x2) break;
      u32 e1 = 0, e2 = 0;
      if (x1 && x2) {
        if (r2r < r1r) e1 = x1;
        else if (r2r == r1r) { e1 = x1; if (x2 != x1) e2 = x2; }
        else e2 = x2;
        bool adv1 = r2r <= r1r, adv2 = r2r >= r1r;
        if (adv1) { x1 = r1p; fresh1 = true; }
        if (adv2) { x2 = r2p; fresh2 = true; }
      } else if (x1) { e1 = x1; x1 = r1p; fresh1 = true; }
      else { e2 = x2; x2 = r2p; fresh2 = true; }
      if (e1) { if (cnt < P.bcap) st(P.off_sync + cnt, e1); else fault |= F_EPOCH_OVERFLOW; cnt++; }
      if (e2) { if (cnt < P.bcap) st(P.off_sync + cnt, e2); else fault |= F_EPOCH_OVERFLOW; cnt++; }
    }
    LBFT_MARK(21);
    if (cnt > P.bcap) cnt = P.bcap;
    u32 epoch = nf(node, NF_EPOCH);
    for (u32 j = cnt; j-- > 0;) {
      u32 b = ld(P.off_sync + j);
      Blk rb = blk_get(b);
      if (rb.epoch() != epoch) continue;  // (cannot happen: a store only holds records of its own epoch)
      insert_block(node, b, rb);
      insert_qc(node, b, rb);
    }
    // A timeout whose round is not the receiver's current round is rejected without side effects, and the current round only
    // moves forward while a set is inserted (see handle_notification): a set of another round is skipped as a whole, the
    // authors the node already holds are not fetched, the others several per round trip.
    const bool pre = LBFT_RESP_FAST && have;  // (the response's own store: these words came with its first burst)
    const u32 tc_raw = pre ? rp.tc_round : ld(base + S_TC_ROUND), to_raw = pre ? rp.to_round : ld(base + S_TO_ROUND);
    const u32 tc_round = wide() ? tc_raw & ~LBFT_S_XFLAG : tc_raw, to_round = wide() ? to_raw & ~LBFT_S_XFLAG : to_raw;
    const bool xft = wide() && (tc_raw & LBFT_S_XFLAG) != 0, xfo = wide() && (to_raw & LBFT_S_XFLAG) != 0;  // (see LBFT_S_XFLAG: clear = the extension words are zero)
    // (the words of a set are fetched together, before the insertions -- see handle_notification)
    if (tc_round == nf(node, NF_CUR_ROUND)) {
      const u32 xb = base + S_FIXED_WORDS + 2 * NN();
      u32 x0 = pre ? rp.tc_mask0 : ld(base + S_TC_MASK), x1 = 0, x2 = 0, x3 = 0;
      if (xft && 1 < MW()) x1 = ld(xb);
      if (xft && 2 < MW()) x2 = ld(xb + 1u);
      if (xft && 3 < MW()) x3 = ld(xb + 2u);
      for (u32 k = 0; k < (xft ? MW() : 1u); k++) {
        if (tc_round != nf(node, NF_CUR_ROUND)) break;
        insert_timeouts_at(node, base + S_FIXED_WORDS, k == 0 ? x0 : k == 1 ? x1 : k == 2 ? x2 : x3, tc_round, 32 * k);
      }
    }
    if (to_round == nf(node, NF_CUR_ROUND)) {
      const u32 xb = base + S_FIXED_WORDS + 2 * NN() + (MW() - 1);
      u32 x0 = pre ? rp.to_mask0 : ld(base + S_TO_MASK), x1 = 0, x2 = 0, x3 = 0;
      if (xfo && 1 < MW()) x1 = ld(xb);
      if (xfo && 2 < MW()) x2 = ld(xb + 1u);
      if (xfo && 3 < MW()) x3 = ld(xb + 2u);
      for (u32 k = 0; k < (xfo ? MW() : 1u); k++) {
        if (to_round != nf(node, NF_CUR_ROUND)) break;
        insert_timeouts_at(node, base + S_FIXED_WORDS + NN(), k == 0 ? x0 : k == 1 ? x1 : k == 2 ? x2 : x3, to_round, 32 * k);
      }
    }
    // the proposed block, unless the burst above found it known to the node already (a bit that is only ever set)
    if (pb && !((pbk >> (node & 31u)) & 1u)) insert_block(node, pb);
  }
  // DataSyncNode::handle_response (data_sync.rs:209-240): `slot` = the peer's store at request time + the request.
  // The reference inserts the unknown records epoch by epoch and runs process_commits + update_tracker between two epochs
  // (:226-236).  handle_response_epoch does ONE epoch -- the first one >= e that is the node's current epoch -- and returns true
  // (e = the next epoch) when that in-between step is due: the event loop runs it at update_node's own process_commits /
  // update_tracker site (update_node(.., tail_only)) instead of a second inlined copy here, which alone cost the
  // two-wavefront kernel 160 spilled registers.
  LBFT_HD bool handle_response_epoch(u32 node, u32 peer, u32 slot, u32& e, const Resp& rp) {
    u32 rbase = sfw(slot, 0);
    u32 req_epoch = rp.req_epoch, req_certs = rp.req_certs;
    u32 peer_epoch = rp.epoch;
    u32 mine = nf(node, NF_EPOCH);
    if (e < mine) e = mine;  // (entries of epochs the node has left are skipped)
    if (e > peer_epoch || e > mine) return false;  // (no entries when the requester was ahead of the peer)
    bool cur = e == peer_epoch;
    u32 base = cur ? rbase : arch_base(peer, e);
    insert_unknown_records(node, base, e == req_epoch, req_certs >> 16, req_certs & 0xffffu, rp, cur);
    if (cur) return false;
    e++;
    return true;
  }
  // The words of a response that decide its walk (the peer's store at request time: epoch, certificates, proposed block) and the
  // request it answers (epoch, certificates), in one burst.
  LBFT_HD Resp load_response(u32 slot) const {
    Resp rp;
    u32 sb = boff(OFFSNAP() + slot * SWORDS());
    rp.epoch = ldf(sb, S_EPOCH); rp.certs = ldf(sb, S_CERTS); rp.prop = ldf(sb, S_PROP_VOTE);
    if (refpack()) rp.epoch &= 0xffffu;
    rp.tc_round = rp.to_round = rp.tc_mask0 = rp.to_mask0 = 0;
    if (LBFT_RESP_FAST) { rp.tc_round = ldf(sb, S_TC_ROUND); rp.to_round = ldf(sb, S_TO_ROUND); rp.tc_mask0 = ldf(sb, S_TC_MASK); rp.to_mask0 = ldf(sb, S_TO_MASK); }
    u32 qb = sqw(sfw(slot, 0), 0);
    rp.req_epoch = ld(qb); rp.req_certs = ld(qb + 1);
    return rp;
  }
  // ---- would handle_response insert anything?  true = provably nothing, decided from the response's first burst, the node's staged words, the node's bits
  // in the "known" / "QC" sets of the (at most two) certificate blocks the walk would start from, the proposed block's "known" word and the sets' extension
  // words; conservative: an archived store of an earlier epoch says false.  No chain walk: a node that holds a block WITH its quorum certificate holds every
  // record of the chain below it (insert_block needs the previous QC, insert_qc the block: record_store.rs:263-291,330-389), so every (block, QC) pair that
  // unknown_records lists below such a head is answered "already inserted".  Together with update_is_noop this holds for 91 % / 99 % / 99 % of the responses
  // of c4live / c5live / c5named, in streaks of 18 / 144 / 125 (tests/tools/run_stats.cpp; by the time a response arrives the notifications have long
  // delivered what the peer's store held at request time) -- coop_responses takes them as runs.  (Round 6's first predicate gave up at any certificate
  // the request had not named: 53 / 48 / 28 %, runs of two.)
  LBFT_HD bool response_is_inert(u32 node, const Resp& rp, u32 slot) const {
    const u32 mine = nf(node, NF_EPOCH);
    u32 e = rp.req_epoch;
    if (e < mine) e = mine;
    if (e > rp.epoch || e > mine) return true;   // (handle_response_epoch: no entries)
    if (e != rp.epoch) return false;             // an archived store of an earlier epoch comes first
    u32 x1 = rp.certs >> 16, x2 = rp.certs & 0xffffu;
    if (e == rp.req_epoch) {
      const u32 k_hqc = rp.req_certs >> 16, k_hcc = rp.req_certs & 0xffffu;
      if (x1 && (x1 == k_hqc || x1 == k_hcc)) x1 = 0;
      if (x2 && (x2 == k_hqc || x2 == k_hcc)) x2 = 0;
    }
    // a certificate whose block the node holds WITH its quorum certificate: every record of the chain below it is held as well (insert_block
    // needs the previous QC, insert_qc the block), so each (block, QC) pair the walk would list returns "already inserted"
    if (x1 && !(bm_known_qc(x1, node))) return false;
    if (x2 && x2 != x1 && !(bm_known_qc(x2, node))) return false;
    const u32 base = sfw(slot, 0), cur = nf(node, NF_CUR_ROUND);
    for (u32 k = 0; k < MW(); k++) {
      const u32 have = am_word(node, NF_TO_MASK, k);
      if ((rp.tc_round & ~LBFT_S_XFLAG) == cur) { u32 tk = k == 0 ? rp.tc_mask0 : ld(base + S_FIXED_WORDS + 2 * NN() + (k - 1)); if (tk & ~have) return false; }
      if ((rp.to_round & ~LBFT_S_XFLAG) == cur) { u32 ok = k == 0 ? rp.to_mask0 : ld(base + S_FIXED_WORDS + 2 * NN() + (MW() - 1) + (k - 1)); if (ok & ~have) return false; }
    }
    const u32 pb = rp.prop & 0xffffu;
    if (pb) {
      const u32 kw = ld((!wide() || node < 32) ? bfw(pb, B_KNOWN) : bxw(pb, B_KNOWN, node >> 5));
      if (!((kw >> (node & 31u)) & 1u)) return false;
    }
    return true;
  }
  // (the node-level interface: the whole call at once)
  LBFT_HD void handle_response(u32 node, u32 peer, u32 slot, i64 lclock) {
    Resp rp = load_response(slot);
    u32 e = rp.req_epoch;
    while (handle_response_epoch(node, peer, slot, e, rp)) update_node(node, lclock, true);
  }

  // `twin`: (E2) the copy for even-indexed receivers of an equivocator's notification carries the twin proposal
  // `skip_hcbr`: the caller copies the timeouts' highest_certified_block_round words itself (coop_bulk: all lanes at once)
  LBFT_HD void write_snapshot(u32 node, u32 slot, bool twin = false, bool skip_hcbr = false) const {
    const u32 sb0 = boff(sfw(slot, 0));  // (record base + immediate field offsets: see bf())
    stf(sb0, S_EPOCH, nf(node, NF_EPOCH));
    // highest_commit_certificate (data_sync.rs:84-92): the current store's, else the previous epoch's store's.
    // Reference quirk Q2: EpochId::previous() returns the SAME epoch (base_types.rs:31-37), so that lookup finds
    // the current store again and yields None; quirks bit 1 makes it id - 1 as intended.
    u32 hcc = nf(node, NF_HCC_BLK);
    if (!hcc && (P.quirks & 2u) && nf(node, NF_EPOCH) != 0) hcc = nf(node, NF_PREV_EPOCH_HCC);
    stf(sb0, S_CERTS, hcc | (nf(node, NF_HQC_BLK) << 16));
    u32 pb = proposed_block(node);
    // "Do not reshare other leaders' proposals."  current_proposed_block is only ever set to a block authored by the
    // leader of its round (record_store.rs:469), and proposed_block() answers for the pacemaker's round: the author is
    // the pacemaker's leader -- no block lookup.
    if (pb && nf(node, NF_PM_LEADER) != node) pb = 0;
    if (twin && pb) pb -= 1;
    u32 vote = 0;  // current_vote(local author) (record_store.rs:762-764)
    if (am_test(node, NF_BAL0_AUTHORS, node)) vote = nf(node, NF_BAL0_BLK);
    else if (am_test(node, NF_BAL1_AUTHORS, node)) vote = nf(node, NF_BAL1_BLK);
    stf(sb0, S_PROP_VOTE, pb | (vote << 16));
    u32 htc = nf(node, NF_HTC_ROUND);
    u32 tcm = htc ? nf(node, NF_TC_MASK) : 0;
    u32 tom = nf(node, NF_TO_MASK);
    u32 xt = 0, xo = 0;
    for (u32 k = 1; wide() && k < MW(); k++) { xt |= htc ? am_word(node, NF_TC_MASK, k) : 0u; xo |= am_word(node, NF_TO_MASK, k); }
    stf(sb0, S_TC_ROUND, htc | (xt ? LBFT_S_XFLAG : 0u));
    stf(sb0, S_TO_ROUND, nf(node, NF_CUR_ROUND) | (xo ? LBFT_S_XFLAG : 0u));
    stf(sb0, S_TC_MASK, tcm);
    stf(sb0, S_TO_MASK, tom);
    u32 tc_sel = nf(node, NF_TC_SEL);
    if (hc_reg()) {
      // both buffers whole, no loop over the sets: a receiver only reads the words of authors in the sets
      LBFT_UNROLL
      for (u32 a = 0; a < 4; a++) if (a < NN()) stf(sb0, S_FIXED_WORDS + a, tc_sel ? hcw[4 + a] : hcw[a]);
      LBFT_UNROLL
      for (u32 a = 0; a < 4; a++) if (a < NN()) stf(sb0, S_FIXED_WORDS + NN() + a, tc_sel ? hcw[a] : hcw[4 + a]);
    } else
    if (!skip_hcbr) {
      copy_hcbr(node, slot, tcm, 0, tc_sel, S_FIXED_WORDS);
      copy_hcbr(node, slot, tom, 0, 1u - tc_sel, S_FIXED_WORDS + NN());
    }
    for (u32 k = 1; wide() && k < MW(); k++) {  // authors >= 32 (n > 32 only): extension words of the two sets + their hcbr entries
      u32 tk = htc ? am_word(node, NF_TC_MASK, k) : 0, ok = am_word(node, NF_TO_MASK, k);
      st(sxw(slot, 0, k), tk);
      st(sxw(slot, 1, k), ok);
      if (!skip_hcbr) {
        copy_hcbr(node, slot, tk, 32 * k, tc_sel, S_FIXED_WORDS);
        copy_hcbr(node, slot, ok, 32 * k, 1u - tc_sel, S_FIXED_WORDS + NN());
      }
    }
  }
  // The hcbr words of a snapshot that write_snapshot(.., skip_hcbr) left out, copied by all lanes of the wavefront at once
  // (lane = author): one load + one store per lane instead of a chain of 8-author batches in the leader lane.  `k` = the
  // leader lane, whose node cache holds the sets.
  LBFT_HD void coop_copy_hcbr(u32 k, u32 l4, u32 node, u32 slot) const { coop_copy_hcbr_to(k, l4, node, sfw(slot, 0)); }
  LBFT_HD void coop_copy_hcbr_to(u32 k, u32 l4, u32 node, u32 base) const {
    const bool is_k = LBFT_IS_LANE(k);
    u32 tw_[4] = {0, 0, 0, 0}, ow_[4] = {0, 0, 0, 0}, sel_k = 0;
    if (is_k) {
      u32 htc = nf(node, NF_HTC_ROUND);
      sel_k = nf(node, NF_TC_SEL);
      LBFT_UNROLL
      for (u32 q = 0; q < 4; q++) { tw_[q] = (q < MW() && htc) ? am_word(node, NF_TC_MASK, q) : 0u; ow_[q] = q < MW() ? am_word(node, NF_TO_MASK, q) : 0u; }
    }
    const u32 tc_sel = LBFT_UNI(sel_k, k);
    u32 tcw[4], tow[4];
    LBFT_UNROLL
    for (u32 q = 0; q < 4; q++) { tcw[q] = LBFT_UNI(tw_[q], k); tow[q] = LBFT_UNI(ow_[q], k); }
    const u32 src_tc = nfw(node, HCO() + tc_sel * NN()), src_to = nfw(node, HCO() + (1u - tc_sel) * NN());
    const u32 dst_tc = base + S_FIXED_WORDS, dst_to = base + S_FIXED_WORDS + NN();
    // (at most two passes of 64 authors: unrolled, so that the set words are picked with compile-time indices -- indexed by the loop
    // variable the two four-word arrays went through scratch memory, a store + a dependent load per pass in every lane of the wavefront)
    LBFT_UNROLL
    for (u32 pass = 0; pass < (LBFT_MAX_NODES + 63) / 64; pass++) {
      const u32 a0 = pass * 64u, q0 = pass * 2u;
      if (a0 >= NN()) break;
      PL<u32> vt, vo, ht, ho;
      LBFT_FOR_LANES(l) {
        u32 wt = l < 32u ? tcw[q0] : tcw[q0 + 1], wo = l < 32u ? tow[q0] : tow[q0 + 1];
        vt[l] = (a0 + l < NN()) ? (wt >> (l & 31u)) & 1u : 0u;
        vo[l] = (a0 + l < NN()) ? (wo >> (l & 31u)) & 1u : 0u;
        ht[l] = vt[l] ? ldc(l4, src_tc + a0 + l) : 0u;
        ho[l] = vo[l] ? ldc(l4, src_to + a0 + l) : 0u;
      }
      LBFT_FOR_LANES(l) {
        if (vt[l]) stc(l4, dst_tc + a0 + l, ht[l]);
        if (vo[l]) stc(l4, dst_to + a0 + l, ho[l]);
      }
    }
  }

  // ---- DataSyncNode::handle_notification (data_sync.rs:113-177); returns should_sync ----
  // The words of a notification that the event loop fetches in ONE burst together with the receiver's node rows: the
  // fixed words, the slot's reference count and (networks of <= 4 nodes) the highest_certified_block_round of the
  // sender's current timeouts -- every later dependent fetch would be a memory round trip of its own, serialised
  // with those of the lanes on other paths.
  LBFT_HD bool small_sets() const { return C0 && NN() <= 4; }
  LBFT_HD Snap load_snapshot(u32 slot) const {
    Snap sn;
    u32 sb = boff(OFFSNAP() + slot * SWORDS());
    LBFT_UNROLL
    for (u32 f = 0; f < S_FIXED_WORDS; f++) sn.w[f] = ldf(sb, f);
    if (refpack()) { sn.refs = sn.w[S_EPOCH] >> 16; sn.w[S_EPOCH] &= 0xffffu; }
    else sn.refs = ld(OFFSREF() + slot);
    LBFT_UNROLL
    for (u32 a = 0; a < 4; a++) sn.to_hcbr[a] = 0;
    if (small_sets()) {
      u32 hb = boff(OFFSNAP() + slot * SWORDS() + S_FIXED_WORDS + NN());
      LBFT_UNROLL
      for (u32 a = 0; a < 4; a++) sn.to_hcbr[a] = ldf(hb, a < NN() ? a : 0);
    }
    return sn;
  }
  LBFT_HD bool handle_notification(u32 node, u32 sender, u32 slot, const Snap& sn) {
    u32 epoch = nf(node, NF_EPOCH);
    u32 n_epoch = sn.w[S_EPOCH];
    bool should_sync = n_epoch > epoch;
    u32 certs = sn.w[S_CERTS];
    u32 hcc = certs & 0xffffu, hqc = certs >> 16;
    // The two certificates of a notification (data_sync.rs:125-148), commit certificate first.  A certificate that IS the
    // receiver's current commit / quorum certificate changes nothing (its QC is in the store, the should_sync tests are
    // false), which is the usual case; what remains is at most one certificate for most lanes, so the two go through
    // one loop: the wavefront pays for max(pending per lane) insertions instead of one per certificate kind.
    const u32 ROLE_HCC = 1u << 31, ROLE_HQC = 1u << 30;
    u32 p0 = 0, p1 = 0;
    if (hcc && hcc != nf(node, NF_HCC_BLK)) p0 = hcc | ROLE_HCC;
    if (hqc && hqc != nf(node, NF_HQC_BLK)) {
      if ((p0 & 0xffffu) == hqc) p0 |= ROLE_HQC;  // same block in both roles
      else if (p0) p1 = hqc | ROLE_HQC;
      else p0 = hqc | ROLE_HQC;
    }
    for (u32 k = 0; k < 2; k++) {
      u32 c = k ? p1 : p0;
      if (!c) break;
      LBFT_STAT(32 + k);
      u32 b = c & 0xffffu;
      Blk r = blk_get(b);
      u32 qe = r.epoch();
      if (qe == epoch) insert_qc(node, b, r);
      if (c & ROLE_HCC) should_sync |= (qe > epoch) || (qe == epoch && r.round() > nf(node, NF_HC_ROUND) + 2);
      if (c & ROLE_HQC) should_sync |= (qe > epoch) || (qe == epoch && r.round() > nf(node, NF_HQC_ROUND));
    }
    LBFT_MARK(20);
    LBFT_MARK(21);
    if (n_epoch == epoch) {
      u32 pv = sn.w[S_PROP_VOTE];
      u32 pb = pv & 0xffffu, vote = pv >> 16;
      if (pb) { LBFT_STAT(34); insert_block(node, pb); }
      LBFT_MARK(22);
      const u32 tc_raw = sn.w[S_TC_ROUND], to_raw = sn.w[S_TO_ROUND];
      u32 tc_round = wide() ? tc_raw & ~LBFT_S_XFLAG : tc_raw, to_round = wide() ? to_raw & ~LBFT_S_XFLAG : to_raw;
      // A timeout whose round is not the receiver's current round is rejected without side effects
      // (record_store.rs:390-415), and the current round only moves forward while a set is inserted: a set whose
      // round differs from the current round on entry is skipped as a whole -- which is the common case, because
      // every notification keeps carrying the sender's last timeout certificate (data_sync.rs:93-96).
      // (large networks: the extension words of a set -- authors >= 32 -- are fetched together, before the insertions whose stores the compiler
      // cannot move a load across: one round trip per set instead of one per word; the slot's words do not change while it is referenced)
      if (tc_round == nf(node, NF_CUR_ROUND)) {
        LBFT_STAT(35);
        u32 x1 = 0, x2 = 0, x3 = 0;
        const bool xf = wide() && (tc_raw & LBFT_S_XFLAG) != 0;  // (clear: the extension words are zero -- not fetched, nothing to insert from them)
        if (xf) { if (1 < MW()) x1 = ld(sxw(slot, 0, 1)); if (2 < MW()) x2 = ld(sxw(slot, 0, 2)); if (3 < MW()) x3 = ld(sxw(slot, 0, 3)); }
        insert_timeouts(node, slot, S_FIXED_WORDS, sn.w[S_TC_MASK], tc_round);
        for (u32 k = 1; xf && k < MW(); k++) insert_timeouts(node, slot, S_FIXED_WORDS, k == 1 ? x1 : k == 2 ? x2 : x3, tc_round, 32 * k);
      }
      if (to_round == nf(node, NF_CUR_ROUND)) {
        if (sn.w[S_TO_MASK]) LBFT_STAT(36);
        u32 x1 = 0, x2 = 0, x3 = 0;
        const bool xf = wide() && (to_raw & LBFT_S_XFLAG) != 0;
        if (xf) { if (1 < MW()) x1 = ld(sxw(slot, 1, 1)); if (2 < MW()) x2 = ld(sxw(slot, 1, 2)); if (3 < MW()) x3 = ld(sxw(slot, 1, 3)); }
        if (small_sets()) {  // hcbr words already fetched with the notification
          for (u32 m = sn.w[S_TO_MASK] & 15u; m; m &= m - 1) {  // (one inlined copy of insert_timeout, not one per author)
            u32 a = ctz32(m);
            u32 h = a == 0 ? sn.to_hcbr[0] : a == 1 ? sn.to_hcbr[1] : a == 2 ? sn.to_hcbr[2] : sn.to_hcbr[3];
            insert_timeout(node, a, to_round, h);
          }
        } else
        insert_timeouts(node, slot, S_FIXED_WORDS + NN(), sn.w[S_TO_MASK], to_round);
        for (u32 k = 1; xf && k < MW(); k++) insert_timeouts(node, slot, S_FIXED_WORDS + NN(), k == 1 ? x1 : k == 2 ? x2 : x3, to_round, 32 * k);
      }
      LBFT_MARK(23);
      if (vote) { LBFT_STAT(37); insert_vote(node, sender, vote, blk_get(vote)); }
      LBFT_MARK(24);
    }
    return should_sync;
  }

  // ---- SimulatedNode::update (simulator.rs:176-179) ----
  LBFT_HD Actions node_update(u32 node, bool tail_only = false) { return update_node(node, (i64)clock - (i64)(i32)nf(node, NF_STARTUP), tail_only); }

  // quirks bit 0: a request carries the requester's epoch and the certificates heading its chains, from which the peer
  // derives known_quorum_certificate_rounds (data_sync.rs:66-71).  Returns a snapshot slot (refcount still 0) or -1.
  LBFT_HD i32 make_request_slot(u32 epoch, u32 certs) {
    i32 rs = snap_alloc();
    if (rs >= 0) { st(sfw((u32)rs, S_EPOCH), epoch); st(sfw((u32)rs, S_CERTS), certs); }
    return rs;
  }

  // ---- receiver / sender lists of process_node_actions (simulator.rs:326-343,356-370) ----
  // n <= 16: sixteen 4-bit entries in one 64-bit register (a dynamically indexed array would be a
  // 32-way select chain per access); larger networks keep the list in a per-instance HBM row region.
  u64 plist;
  u32 plist8;     // n <= 8 in kernel class 0: eight 4-bit entries (64-bit variable shifts cost three instructions each)
  u8* plist_lds;  // n > 16: this instance's 128-byte list in LDS (device); nullptr = the HBM row region
  LBFT_HD bool packed8() const { return C0 && NN() <= 8; }
  LBFT_HD u32 peer(u32 i) const {
    if (packed8()) return (plist8 >> (4 * i)) & 15u;
    if (packed()) return (u32)(plist >> (4 * i)) & 15u;
    if (plist_lds) return plist_lds[i];
    return ld(P.off_list + i);
  }
  LBFT_HD void peers_one(u32 a) {
    if (packed8()) plist8 = a; else
    if (packed()) plist = a; else if (plist_lds) plist_lds[0] = (u8)a; else st(P.off_list, a);
  }
  LBFT_HD u32 peers_all_but(u32 node) {  // all other nodes in index order
    if (packed8()) {
      const u32 ident = 0x76543210u;
      u32 low = ident & ((1u << (4 * node)) - 1u);
      u32 high = node < 7 ? ((ident >> (4 * (node + 1))) << (4 * node)) : 0;
      plist8 = low | high;
    } else if (packed()) {
      const u64 ident = 0xfedcba9876543210ULL;
      u64 low = node ? (ident & ((1ULL << (4 * node)) - 1)) : 0;
      u64 high = node < 15 ? ((ident >> (4 * (node + 1))) << (4 * node)) : 0;
      plist = low | high;
    } else if (plist_lds) {
      u32 c = 0;
      for (u32 i = 0; i < NN(); i++) if (i != node) plist_lds[c++] = (u8)i;
    } else {
      u32 c = 0;
      for (u32 i = 0; i < NN(); i++) if (i != node) st(P.off_list + c++, i);
    }
    return NN() - 1;
  }
  LBFT_HD void peers_shuffle(u32 cnt) {  // rand 0.8 SliceRandom::shuffle (no draws when cnt < 2)
    for (u32 i = cnt; i-- > 1;) {
      u32 j = rng.gen_range_u32(i + 1);
      if (packed8()) {
        u32 x = ((plist8 >> (4 * i)) ^ (plist8 >> (4 * j))) & 15u;
        plist8 ^= (x << (4 * i)) | (x << (4 * j));
      } else if (packed()) {
        u64 x = ((plist >> (4 * i)) ^ (plist >> (4 * j))) & 15ULL;
        plist ^= (x << (4 * i)) | (x << (4 * j));
      } else if (plist_lds) {
        u8 a = plist_lds[i], b = plist_lds[j];
        plist_lds[i] = b; plist_lds[j] = a;
      } else {
        u32 a = ld(P.off_list + i), b = ld(P.off_list + j);
        st(P.off_list + i, b); st(P.off_list + j, a);
      }
    }
  }

  // ---- Simulator::process_node_actions (simulator.rs:296-378) ----
  LBFT_HD void process_node_actions(u32 node, const Actions& act) {
    i64 startup = (i64)(i32)nf(node, NF_STARTUP);
    i64 t_new = (i64)((u64)act.next + (u64)startup);
    if (t_new < (i64)clock + 1) t_new = (i64)clock + 1;
    i64 ign = t_new - 1;
    if (ign > (i64)P.max_clock) ign = P.max_clock;  // only ever compared with clock <= max_clock
    nfs(node, NF_IGNORE_UNTIL, (u32)(i32)ign);
    // Duplicate-timer folding.  A second UpdateTimerEvent for the same (node, time) is always a no-op in
    // the reference: all timers of one time pop consecutively (kind 3 sorts first, simulator.rs:149-161)
    // and the first one either fires -- which moves ignore_scheduled_updates_until to >= clock -- or is
    // itself cancelled, and then so are the others.  So if the node's previously scheduled timer has the
    // same time (it is still pending: that time is > clock), only count the duplicate.
    if (t_new <= (i64)P.max_clock && (u32)t_new == nf(node, NF_LAST_TIMER_T)) {
      LBFT_STAT(40);
#if !defined(LBFT_NO_EXEC_COUNTERS)
      n_fold++;
#endif
      nfs(node, NF_TIMER_DUPS, nf(node, NF_TIMER_DUPS) + 1);
      nfs(node, NF_DUP_STAMP, stamp);
      stamp++;
    } else {
      if (t_new <= (i64)P.max_clock) {
        // The tracked timer time moves on.  When the round-switch trace is on (it needs every pop at its own time;
        // otherwise they are simply counted with the next tracked timer of this node, which leaves all event totals
        // unchanged), duplicates still folded into the previous one are materialised as
        // ONE counted no-op event at their own time, carrying the creation stamp of the last of them (it pops
        // after its original, hence cancelled), so that event counts and the round-switch trace stay attributed
        // to the right time and queue position.
        u32 dups = nf(node, NF_TIMER_DUPS);
        if (dups && tracing()) {
          push_event((i64)(i32)nf(node, NF_LAST_TIMER_T), 3, node, 0, dups - 1, nf(node, NF_DUP_STAMP));
          nfs(node, NF_TIMER_DUPS, 0);
        }
        nfs(node, NF_LAST_TIMER_T, (u32)t_new);
      }
      LBFT_STAT(41); if (t_new > (i64)P.max_clock) LBFT_STAT(42);
      push_event(t_new, 3, node, 0, 0);
    }
    LBFT_MARK(12);
  }

  // ---- every network message of one event, in the reference's order and with its RNG draws, through ONE loop ----
  // simulator.rs: the response to a request (:441-453); or, after update_node, the sync request of a notification
  // (:424-433), the notifications to the shuffled receivers (:326-354) and the requests to the shuffled senders
  // (:356-377).  The reference has four send sites; lanes of a wavefront usually need different ones, so separate sites
  // would each run (delay sampler included) for a few lanes.  Here a lane's messages form one list and iteration j of
  // the loop sends every lane's j-th message: the wavefront pays max(messages per lane) sampler runs instead of one
  // per site and loop iteration.
  struct SendPlan {
    u32 response;      // 1: this event is a request; answer it (slot = response snapshot under quirks bit 0)
    u32 resp_slot;
    u32 sync;          // 1: handle_notification asked for a request to the notification's sender
    u32 sync_stamp;    // its creation stamp, reserved before the timer was scheduled (order of simulator.rs:424-440)
    u32 sync_epoch, sync_certs;
    u32 have_actions;  // update_node ran: act is valid
  };
  LBFT_HD void send_loop(u32 node, u32 sender, const SendPlan& sp, const Actions& act) {
    u32 n_a = 0, n_b = 0;
    if (sp.have_actions) {
      if (act.broadcast) n_a = NN() - 1;
      else if (act.send_to >= 0 && (u32)act.send_to != node) n_a = 1;
      if (act.query_all) n_b = NN() - 1;
    }
    u32 first_a = sp.response + sp.sync, first_b = first_a + n_a, total = first_b + n_b;
    // Cooperative kernels: a list of n - 1 messages (broadcast / query-all) is left to coop_bulk, which run_coop executes with
    // all lanes of the wavefront right after this loop -- same order, same draws; what precedes the lists (response, sync
    // request, a single notification) is sent here.  (n > 32: n_a is 0, 1 or n - 1 and n_b is 0 or n - 1.)
    bulk = 0;
    if (coop()) {
      if (n_a > 1) bulk |= 1u;
      if (n_b > 0) bulk |= 2u;
      total = first_a + (n_a == 1 ? 1u : 0u);
      bulk_node = node;
    }
    LBFT_STAT(16 + (total > 7 ? 7 : total)); LBFT_STATN(24, total);
    if (sp.have_actions) { if (act.broadcast) LBFT_STAT(25); else if (n_a) LBFT_STAT(26); if (act.query_all) LBFT_STAT(27); if (sp.sync) LBFT_STAT(28); }
    i32 slot = -1, slot_twin = -1, rs = 0;
    u32 refs = 0, refs_twin = 0, rrefs = 0;
    bool equivocal = false;
    for (u32 j = 0; j < total; j++) {
      // receivers.shuffle(rng) (simulator.rs:343) / create_request + senders.shuffle(rng) (simulator.rs:365-370): each
      // drawn right before the delays of its list; one site for both lists (they never start at the same j)
      bool start_a = j == first_a && n_a != 0, start_b = j == first_b && n_b != 0;
      if (start_a || start_b) {
        if (start_b || act.broadcast) peers_all_but(node); else peers_one((u32)act.send_to);
        if (start_b) rs = q1() ? make_request_slot(nf(node, NF_EPOCH), nf(node, NF_HCC_BLK) | (nf(node, NF_HQC_BLK) << 16)) : 0;
        peers_shuffle(start_a ? n_a : n_b);
        if (start_a && is_equivocator(node)) {  // (E2): does this notification carry one of the node's own (double) proposals?
          u32 pb = proposed_block(node);
          equivocal = pb != 0 && blk_get(pb).author() == node;
        }
      }
      LBFT_MARK(16);
      i64 t = sample_delay();
      t = t > INT64_MAX - (i64)clock ? INT64_MAX : (i64)clock + t;  // (a saturated sample must not wrap; it is past every horizon anyway)
      LBFT_MARK(18);
      // Each kind of message only decides WHAT is scheduled; the event itself is pushed at one site below (four inlined
      // copies of push_event, one per kind, would each run for the few lanes that need it).
      // mode: 0 = nothing is scheduled but the creation stamp is consumed, 1 = push, 2 = nothing at all
      u32 mode, which, pk, pto, pfrom, pslot = 0, preuse = ~0u;
      if (j < sp.response) {  // DataSyncResponseEvent back to the requester
        which = 0; pk = 2; pto = node; pfrom = sender;
        bool lost = net_lost(node, sender);
        if (q1()) { i32 rslot = (i32)sp.resp_slot; mode = (lost || rslot < 0) ? 0u : 1u; pslot = (u32)rslot; }  // -1: no slot was available
        else mode = lost ? 0u : 1u;
      } else if (j < first_a) {  // the sync request (its stamp was reserved before the timer's)
        which = 1; pk = 1; pto = node; pfrom = sender; preuse = sp.sync_stamp;
        i32 qs = q1() ? make_request_slot(sp.sync_epoch, sp.sync_certs) : 0;
        mode = (net_lost(node, sender) || qs < 0) ? 2u : 1u;
        pslot = (u32)qs;
      } else if (j < first_b) {  // DataSyncNotifyEvent to receiver r
        u32 r = peer(j - first_a);
        pk = 0; pto = r; pfrom = node; which = 2;
        if (net_lost(node, r)) mode = 0;  // a lost message still consumes its creation stamp
        else if (equivocal && (r & 1u) == 0) {
          which = 3;
          if (t <= (i64)P.max_clock && slot_twin == -1) {
            slot_twin = snap_alloc();
            if (slot_twin < 0) slot_twin = -2; else write_snapshot(node, (u32)slot_twin, true);
          }
          mode = slot_twin >= 0 ? 1u : 0u; pslot = (u32)slot_twin;
        } else {
          if (t <= (i64)P.max_clock && slot == -1) {
            slot = snap_alloc();
            if (slot < 0) slot = -2; else write_snapshot(node, (u32)slot);
          }
          mode = slot >= 0 ? 1u : 0u; pslot = (u32)slot;  // a dropped event still consumes a creation stamp
        }
      } else {  // DataSyncRequestEvent to sender s (query_all)
        u32 sd = peer(j - first_b);
        which = 4; pk = 1; pto = node; pfrom = sd; pslot = (u32)rs;
        mode = (net_lost(node, sd) || rs < 0) ? 0u : 1u;
      }
      bool pushed = false;
      if (mode == 1) pushed = push_event(t, pk, pto, pfrom, pslot, preuse);
      else if (mode == 0) stamp++;
      if (which == 2) refs += pushed ? 1u : 0u;
      else if (which == 3) refs_twin += pushed ? 1u : 0u;
      else if (which == 4) rrefs += pushed ? 1u : 0u;
      else if (q1() && (i32)pslot >= 0) {  // response / sync request under quirks bit 0: the slot travels with the event
        if (pushed) snap_set_refs(pslot, 1, which == 0 ? nf(node, NF_EPOCH) /* the peer's store: it is the node in the cache of this step */ : sp.sync_epoch);
        else snap_free_slot(pslot);
      }
      LBFT_MARK(19);
    }
    if (slot >= 0) { if (refs) snap_set_refs((u32)slot, refs, nf(node, NF_EPOCH)); else snap_free_slot((u32)slot); }
    if (slot_twin >= 0) { if (refs_twin) snap_set_refs((u32)slot_twin, refs_twin, nf(node, NF_EPOCH)); else snap_free_slot((u32)slot_twin); }
    if (q1() && n_b && rs >= 0 && !(bulk & 2u)) { if (rrefs) snap_set_refs((u32)rs, rrefs, nf(node, NF_EPOCH)); else snap_free_slot((u32)rs); }
    LBFT_MARK(13);
  }

  // ---- cooperative bulk send: the n - 1 messages of one broadcast (which = 0: DataSyncNotify to the shuffled receivers,
  // simulator.rs:326-354) or one query-all (which = 1: DataSyncRequest to the shuffled senders, :356-377) of the network in
  // lane `k`, executed by ALL lanes of the wavefront: lanes = messages.  The reference's order is kept exactly -- shuffle
  // draws, then one delay sample per receiver in shuffled order, creation stamps in that order, FIFO order inside every
  // (time, kind) bucket of the calendar queue -- but only what is inherently serial runs serially:
  //   * the Fisher-Yates shuffle: a wavefront-uniform loop over draws that all lanes fetched from the ring with one load;
  //   * the delay samples: every lane evaluates one ring draw as if a sample started there (ziggurat first try +
  //     exp, or the uniform model's first try); a ballot of "first try accepted" tells how many consecutive samples those
  //     lanes settle at once, and only a draw that needs the rejection path (~1.2 %) is sampled serially by the leader;
  //   * the pushes: slots come from the stack of freed slots / the bump allocator by rank (independent loads), lanes whose
  //     messages fall into the same bucket find each other with bit-sliced ballots over the time, link themselves in lane
  //     (= stamp) order, and one lane per bucket appends the chain to the bucket's tail -- one dependent load per distinct
  //     bucket instead of one memory round trip per message.
  // ---- lanes = scheduled times: ONE delay sample per message of a list of `cnt` (<= 128), in list order, off the ring of pre-generated draws of the
  // network in lane k.  Every lane evaluates one ring draw as if a sample started there; a ballot of "first try accepted" tells how many consecutive
  // samples settle at once, a draw that needs the rejection path (~1.2 %) is sampled serially by the leader.  tm0 / tm1: lane j (j - 64) = scheduled
  // time of message j, 0xffffffff = past the horizon.  `stop_zero`: stop behind the first message whose delay is zero (the caller must not run ahead of
  // an event scheduled AT the current time: coop_requests); returns the number of samples taken.
  LBFT_HD u32 coop_sample(u32 k, u32 l4, i32 clk, u32 cnt, PL<u32>& tm0, PL<u32>& tm1, bool stop_zero) {
    const bool is_k = LBFT_IS_LANE(k);
    const u32 ring_row = P.off_ring, ring_mask = P.ring - 1u;
    LBFT_FOR_LANES(l) { tm0[l] = 0xffffffffu; tm1[l] = 0xffffffffu; }
    u32 j = 0;
    while (j < cnt) {
      u32 avail = LBFT_UNI(rng.rcnt, k);
      if (avail == 0) {
        if (is_k) rng.ring_fill(rng.ring_room() < 64u ? rng.ring_room() : 64u);
        avail = LBFT_UNI(rng.rcnt, k);
      }
      const u32 take = avail < 64u ? avail : 64u;
      const u32 head = LBFT_UNI(rng.rhead, k);
      PL<u32> fast, tv, zero;
      LBFT_FOR_LANES(l) {
        fast[l] = 0; tv[l] = 0xffffffffu; zero[l] = 0;
        if (l < take) {
          u32 e = ring_row + 2u * ((head + l) & ring_mask);
          u64 bits = (u64)ldc(l4, e) | ((u64)ldc(l4, e + 1u) << 32);
          i64 d;
          fast[l] = fast_delay(bits, d) ? 1u : 0u;
          tv[l] = horizon_time(clk, d);
          zero[l] = d == 0 ? 1u : 0u;
        }
      }
      const u64 F = pl_ballot(fast);
      u32 run = ~F ? ctz64(~F) : 64u;  // consecutive first-try samples from the head of the ring
      if (run > take) run = take;
      if (run > cnt - j) run = cnt - j;
      bool stop = false;
      if (stop_zero && run) {
        const u64 Z = pl_ballot(zero) & (run >= 64u ? ~0ULL : ((1ULL << run) - 1ULL));
        if (Z) { run = ctz64(Z) + 1u; stop = true; }
      }
      {  // message j + q takes lane q's value (q < run)
        PL<u32> src, got;
        LBFT_FOR_LANES(l) src[l] = (l - j) & 63u;
        pl_shfl(got, tv, src);
        LBFT_FOR_LANES(l) if (l >= j && l < j + run) tm0[l] = got[l];
        if (cnt > 64u) {
          LBFT_FOR_LANES(l) src[l] = (64u + l - j) & 63u;
          pl_shfl(got, tv, src);
          LBFT_FOR_LANES(l) if (64u + l >= j && 64u + l < j + run) tm1[l] = got[l];
        }
      }
      if (is_k) { rng.rhead += run; rng.rcnt -= run; rng.draws += run; }
      j += run;
      if (stop) break;
      if (j < cnt && run < take) {  // the draw at the head needs the rejection path: the leader samples it serially
        u32 tvk = 0, zk = 0;
        if (is_k) { i64 d = sample_delay(); tvk = horizon_time(clk, d); zk = d == 0 ? 1u : 0u; }
        u32 tvu = LBFT_UNI(tvk, k);
        if (j < 64u) pl_write(tm0, j, tvu); else pl_write(tm1, j - 64u, tvu);
        j++;
        if (stop_zero && LBFT_UNI(zk, k)) break;
      }
    }
    return j;
  }
  // ---- lanes = messages of ONE kind for the network in lane k, in lane (= creation stamp) order: `live` lanes are appended to the calendar buckets
  // (t[l], kc) with the event word meta[l].  Returns the ballot of the lanes that were scheduled (a full queue drops the surplus and raises the fault).
  LBFT_HD u64 bulk_append(u32 k, u32 l4, i32 clk, u32 kc, PL<u32>& live, const PL<u32>& t, const PL<u32>& meta) {
    const bool is_k = LBFT_IS_LANE(k);
      u64 L = pl_ballot(live);
      u32 nl = popc64(L);
      const u32 qlen_u = LBFT_UNI(qlen, k);
      const u32 room = P.qcap > qlen_u ? P.qcap - qlen_u : 0u;
      if (nl > room) {  // queue overflow: the surplus is not scheduled
        LBFT_FOR_LANES(l) if (live[l] && popc64(L & ((1ULL << l) - 1ULL)) >= room) live[l] = 0;
        if (is_k) fault |= F_QUEUE_OVERFLOW;
        L = pl_ballot(live);
        nl = popc64(L);
      }
      if (nl) {
        const u32 fc = LBFT_UNI(cal_free, k), bump = LBFT_UNI(cal_bump, k);
        // lanes whose messages share a bucket (same time; the kind is common): M; sharing a bitmap word (same time >> 3): M3
        PL<u64> M, M3;
        LBFT_FOR_LANES(l) { M[l] = live[l] ? L : 0; M3[l] = M[l]; }
        for (u32 b = 0; (1u << b) <= (u32)P.max_clock; b++) {
          PL<u32> bit;
          LBFT_FOR_LANES(l) bit[l] = (live[l] && ((t[l] >> b) & 1u)) ? 1u : 0u;
          const u64 B = pl_ballot(bit);
          if (B == 0 || B == L) continue;
          LBFT_FOR_LANES(l) { u64 m = bit[l] ? B : ~B; M[l] &= m; if (b >= 3u) M3[l] &= m; }
        }
        LBFT_CMARK(9);  // grouping
        // A group = the messages of one bucket, in lane (= stamp) order.  Its first lane (the leader) reads the bucket's tail word; member number rk
        // takes entry base + rk behind the tail chunk's last one, spilling into up to three new chunks, which the leaders take from the stack of freed
        // chunks / the bump allocator by rank (a two-bit-plane ballot prefix over the leaders' needs): one dependent load per distinct bucket.
        PL<u32> lead, rk, gs, tl;
        LBFT_FOR_LANES(l) {
          lead[l] = live[l] ? ctz64(M[l]) : l;
          rk[l] = popc64(M[l] & ((1ULL << l) - 1ULL));
          gs[l] = popc64(M[l]);
          tl[l] = (live[l] && rk[l] == 0) ? ldc(l4, calt(t[l] * 4u + kc)) : 0u;
        }
        PL<u32> tg;  // the group's tail word, in every member
        pl_shfl(tg, tl, lead);
        PL<u32> qn, q0, q1b;  // new chunks a leader needs (0..3)
        LBFT_FOR_LANES(l) {
          qn[l] = 0;
          if (live[l] && rk[l] == 0) { u32 base = tg[l] ? (tg[l] & 63u) : 0u; qn[l] = (base + gs[l] + LBFT_CAL_CE - 1u) / LBFT_CAL_CE - (tg[l] ? 1u : 0u); }
          q0[l] = qn[l] & 1u; q1b[l] = (qn[l] >> 1) & 1u;
        }
        const u64 Q0 = pl_ballot(q0), Q1 = pl_ballot(q1b);
        const u32 total_new = popc64(Q0) + 2u * popc64(Q1);
        const u32 pool = P.cal_chunks;
        PL<u32> id0, id1, id2;
        LBFT_FOR_LANES(l) {
          const u64 lower = (1ULL << l) - 1ULL;
          const u32 a0 = popc64(Q0 & lower) + 2u * popc64(Q1 & lower);
          u32 ids[3] = {0, 0, 0};
          LBFT_UNROLL
          for (u32 j = 0; j < 3; j++)
            if (j < qn[l]) {
              u32 ar = a0 + j;  // allocation rank of this chunk
              u32 c = ar < fc ? ldc(l4, P.off_qlo + fc - 1u - ar) : bump + (ar - fc);
              ids[j] = c < pool ? c : pool - 1u;  // (pool exhausted: the leader lane raises the fault below; stay in bounds)
            }
          id0[l] = ids[0]; id1[l] = ids[1]; id2[l] = ids[2];
        }
        PL<u32> g0, g1, g2;
        pl_shfl(g0, id0, lead); pl_shfl(g1, id1, lead); pl_shfl(g2, id2, lead);
        LBFT_CMARK(8);  // chunks
        PL<u32> needbit;
        LBFT_FOR_LANES(l) {
          needbit[l] = 0;
          if (live[l]) {
            const u32 have = tg[l] ? 1u : 0u, base = tg[l] ? (tg[l] & 63u) : 0u;
            const u32 e = base + rk[l], q = e / LBFT_CAL_CE, pos = e % LBFT_CAL_CE + 1u;
            const u32 c = q < have ? (tg[l] >> 6) - 1u : (q - have == 0 ? g0[l] : q - have == 1 ? g1[l] : g2[l]);
            stc(l4, chw(c, pos), meta[l]);
            if (rk[l] == 0) {  // the leader: links between the chunks, the bucket's tail (and head) word
              const u32 idx = t[l] * 4u + kc, n_new = qn[l];
              if (have && n_new) stc(l4, chw((tg[l] >> 6) - 1u, 0), g0[l] + 1u);
              if (n_new > 1u) stc(l4, chw(g0[l], 0), g1[l] + 1u);
              if (n_new > 2u) stc(l4, chw(g1[l], 0), g2[l] + 1u);
              const u32 lastc = n_new == 0 ? (tg[l] >> 6) - 1u : n_new == 1 ? g0[l] : n_new == 2 ? g1[l] : g2[l];
              const u32 last_e = base + gs[l] - 1u;
              stc(l4, calt(idx), ((lastc + 1u) << 6) | (last_e % LBFT_CAL_CE + 1u));
              if (!have) { stc(l4, calh(idx), ((g0[l] + 1u) << 6) | 1u); needbit[l] = 1; }
            }
          }
        }
        LBFT_CMARK(10);  // entries, links, heads, tails
        const u64 NB = pl_ballot(needbit);
        if (NB) {  // occupancy bits of buckets that were empty: one read-modify-write per bitmap word
          PL<u32> bits;
          LBFT_FOR_LANES(l) bits[l] = 0;
          for (u32 v = 0; v < 8u; v++) {
            PL<u32> is_v;
            LBFT_FOR_LANES(l) is_v[l] = (needbit[l] && (t[l] & 7u) == v) ? 1u : 0u;
            const u64 Bv = pl_ballot(is_v);
            if (!Bv) continue;
            LBFT_FOR_LANES(l) if (needbit[l] && (M3[l] & Bv)) bits[l] |= 1u << (v * 4u + kc);
          }
          LBFT_FOR_LANES(l) {
            if (needbit[l] && (M3[l] & NB & ((1ULL << l) - 1ULL)) == 0) {
              u32 w = P.off_cal_bm + (t[l] >> 3);
              stc(l4, w, ldc(l4, w) | bits[l]);
            }
          }
        }
        if (is_k) {
          qlen += nl;
          if (qlen > maxq) maxq = qlen;
          cal_free = fc - (total_new < fc ? total_new : fc);
          cal_bump = bump + (total_new > fc ? total_new - fc : 0u);
          if (cal_bump > pool) { fault |= F_QUEUE_OVERFLOW; cal_bump = pool; }
          u32 lo = (u32)clk * 4u + kc;  // every message lies at or after the current time: a lower bound for the cursor
          if (lo < cal_cursor) cal_cursor = lo;
        }
      }
    return L;
  }
  // ---- cooperative request run (round 6; the bucket-parallel step where it is exact): under the reference's quirk Q1 a DataSyncRequest is answered by the
  // requester itself with a payload-free response (simulator.rs:441-453) -- the event reads and writes no node, it draws ONE delay and schedules the
  // response.  The requests of one (time, kind) bucket are therefore independent of one another; only their draws and creation stamps are ordered.  The
  // chunked calendar hands a wavefront up to 31 consecutive requests of the network in lane k in one line: lanes = requests -- one load each for the
  // entries, the delay samples off the draw ring by ballot (coop_sample), ONE grouped append of the responses (bulk_append) -- instead of one event-loop
  // step per request (a third of all events of a 64- / 100-node network).  A response scheduled AT the current time pops before the bucket's remaining
  // requests (ScheduledEvent::cmp: responses sort before requests): the run stops behind it (coop_sample(.., stop_zero)).
  u32 req_done;  // leader lane: requests the last coop_requests consumed
  // quirks bit 0 (requests answered by the PEER, handle_request data_sync.rs:183-207): handling a request reads the peer's store and writes nothing of any
  // node, so the requests of a run stay independent -- lanes = requests, each lane stages the peer of ITS request (the node cache of a lane is dead between
  // two events, and for the length of this call every lane addresses the column of the network in lane k) and writes the response snapshot of its own.
  // What is shared is resolved by ballots: the request slots (the n - 1 requests of a query-all share one: the first lane of each group of equal slots takes
  // the group's references off at once), the stack of free slots (released slots are pushed, response slots popped, by rank).  Lanes whose response
  // falls behind the horizon take no slot (event by event the slot would be taken and handed back: nothing of it remains); `live` loses the lanes that
  // find the stack empty, meta[l] gets the response's slot.
  LBFT_HD void coop_answer_requests(u32 k, u32 l4, u32 cnt, PL<u32>& live, PL<u32>& meta) {
    const bool is_k = LBFT_IS_LANE(k);
    const u32 own_l4 = lane4;
    lane4 = l4;
    // the requests' slots: epoch | references << 16, certificates
    PL<u32> rslot, ew, rcerts, inrun;
    LBFT_FOR_LANES(l) {
      inrun[l] = l < cnt ? 1u : 0u;
      rslot[l] = meta[l] >> 16; ew[l] = 0; rcerts[l] = 0;
      if (inrun[l]) { u32 qb = sfw(rslot[l], 0); ew[l] = ld(qb + S_EPOCH); rcerts[l] = ld(qb + S_CERTS); }
    }
    const u64 RUN = pl_ballot(inrun);
    PL<u64> MS;  // lanes of the run whose requests share this lane's slot
    LBFT_FOR_LANES(l) MS[l] = inrun[l] ? RUN : 0;
    for (u32 b = 0; (1u << b) < P.scap; b++) {
      PL<u32> bit;
      LBFT_FOR_LANES(l) bit[l] = (inrun[l] && ((rslot[l] >> b) & 1u)) ? 1u : 0u;
      const u64 B = pl_ballot(bit);
      if (B == 0 || B == RUN) continue;
      LBFT_FOR_LANES(l) MS[l] &= bit[l] ? B : ~B;
    }
    PL<u32> freed;
    LBFT_FOR_LANES(l) {
      freed[l] = 0;
      if (inrun[l] && (MS[l] & ((1ULL << l) - 1ULL)) == 0) {  // first lane of its group
        u32 left = (ew[l] >> 16) - popc64(MS[l]);
        st(sfw(rslot[l], S_EPOCH), (ew[l] & 0xffffu) | (left << 16));
        freed[l] = left == 0 ? 1u : 0u;
      }
    }
    const u64 FR = pl_ballot(freed);
    const u32 sf0 = LBFT_UNI(snap_free, k);
    LBFT_FOR_LANES(l) if (freed[l]) st(OFFSFREE() + sf0 + popc64(FR & ((1ULL << l) - 1ULL)), rslot[l]);
    const u32 sf1 = sf0 + popc64(FR);
    // response slots for the lanes whose response will be scheduled
    const u64 A = pl_ballot(live);
    PL<u32> rs;
    LBFT_FOR_LANES(l) {
      rs[l] = 0;
      if (live[l]) {
        u32 rank = popc64(A & ((1ULL << l) - 1ULL));
        if (rank < sf1) rs[l] = ld(OFFSFREE() + sf1 - 1u - rank); else live[l] = 0;  // (no slot left: not scheduled, the stamp is consumed)
      }
    }
    const u32 want = popc64(A), got = want < sf1 ? want : sf1;
    if (is_k) {
      if (got < want) fault |= F_SNAP_OVERFLOW;
      snap_free = sf1 - got;
      u32 in_use = P.scap - snap_free;
      if (in_use > maxsnap) maxsnap = in_use;
    }
    // the peer's store at request time + the request it answers (handle_request); one reference: the response event
    LBFT_FOR_LANES(l) {
      if (live[l]) {
        const u32 peer = (meta[l] >> 8) & 0xffu;
        begin_node(peer);
        const u32 rb = sfw(rs[l], 0);
        write_store_snapshot(peer, rb, false);
        st(sqw(rb, 0), ew[l] & 0xffffu); st(sqw(rb, 1), rcerts[l]);
        st(rb + S_EPOCH, nf(peer, NF_EPOCH) | (1u << 16));
        meta[l] = (meta[l] & 0xffffu) | (rs[l] << 16);
      }
    }
    lane4 = own_l4;
  }
  LBFT_HD void coop_requests(u32 k, u32 budget) {
    const bool is_k = LBFT_IS_LANE(k);
    const u32 l4 = LBFT_UNI(lane4, k);
    const u32 h = LBFT_UNI(cur_h, k), tailw = LBFT_UNI(sp_nx, k), idx = LBFT_UNI(sp_idx, k);
    const u32 c = (h >> 6) - 1u, pos = h & 63u;
    u32 avail = ((tailw >> 6) == (h >> 6) ? (tailw & 63u) : LBFT_CAL_CE) - pos + 1u;
    if (avail > budget) avail = budget;
    if (is_k && (i32)(idx >> 2) > clock) clock = (i32)(idx >> 2);
    const i32 clk = (i32)LBFT_UNI((u32)clock, k);
    PL<u32> meta;
    LBFT_FOR_LANES(l) meta[l] = l < avail ? ldc(l4, chw(c, pos + l)) : 0u;
    PL<u32> tm0, tm1;
    const u32 cnt = coop_sample(k, l4, clk, avail, tm0, tm1, true);
    PL<u32> live, zero;
    LBFT_FOR_LANES(l) { live[l] = (l < cnt && tm0[l] != 0xffffffffu) ? 1u : 0u; zero[l] = (live[l] && tm0[l] == (u32)clk) ? 1u : 0u; }
    const u64 Z = pl_ballot(zero);
    if (q1()) coop_answer_requests(k, l4, cnt, live, meta);  // quirks bit 0: the peers' stores travel with the responses
    if (is_k) qlen -= cnt;  // (the pops first: the queue's high-water mark is what the event-by-event order reaches)
    bulk_append(k, l4, clk, 1u /* 3 - DataSyncResponse */, live, tm0, meta);  // the response carries the request's (node, sender): same event word
    if (is_k) {
      ev1 += cnt;
      stamp += cnt;
      if (stamp >= (1u << 30)) fault |= F_STAMP_OVERFLOW;
      if (!Z) cal_cursor = idx;  // (bulk_append lowered the cursor to the responses' kind at this time: nothing was scheduled there)
      cal_advance(cnt);
      req_done = cnt;
    }
    LBFT_STAT(30); LBFT_STATN(31, cnt);
    LBFT_FOR_LANES(l) if (l < cnt) LBFT_HOOK_POP(clk, 1u, meta[l] & 0xffu, (meta[l] >> 8) & 0xffu);  // (host analysis tools: the run's events are pops too)
    LBFT_MARK(4);  // (diagnostic builds: a request run is charged to the requests' phase)
  }
  // ---- cooperative response run (round 6), reference semantics (quirk Q1): a DataSyncResponse carries nothing insertable (simulator.rs:454-466 on a response the
  // requester built from its own store), so the event is update_node + the timer of process_node_actions, and on a settled node -- every other event left it
  // right behind an update_node -- that update changes nothing (update_is_noop).  The responses at the head of the open bucket are taken by the whole
  // wavefront, lanes = events: each lane stages the node of ITS event (for the length of this call every lane addresses the column of the network in lane k;
  // a lane's node cache is dead between two events), evaluates the predicate and the timer it schedules; the run ends in front of the first event whose
  // update would do anything -- that one takes the ordinary step.  Events of the run for the SAME node are a group: event by event the first schedules (or
  // folds) the timer, every later one folds into it (same node state, same time: process_node_actions' duplicate-timer rule); the group's first lane writes
  // the node's four timer words once.  Creation stamps by lane, timers appended in lane order (bulk_append), counters by ballot.
  u32 rsp_done;  // leader lane: responses the last coop_responses consumed (0: the head event is not a no-op)
  LBFT_HD void coop_responses(u32 k, u32 budget) {
    const bool is_k = LBFT_IS_LANE(k);
    const u32 l4 = LBFT_UNI(lane4, k);
    const u32 h = LBFT_UNI(cur_h, k), tailw = LBFT_UNI(sp_nx, k), idx = LBFT_UNI(sp_idx, k);
    const u32 c = (h >> 6) - 1u, pos = h & 63u;
    u32 avail = ((tailw >> 6) == (h >> 6) ? (tailw & 63u) : LBFT_CAL_CE) - pos + 1u;
    if (avail > budget) avail = budget;
    if (is_k && (i32)(idx >> 2) > clock) clock = (i32)(idx >> 2);
    const i32 clk = (i32)LBFT_UNI((u32)clock, k);
    const u32 stamp0 = LBFT_UNI(stamp, k);
    const u32 sf0 = RSPRUNQ ? LBFT_UNI(snap_free, k) : 0u;
    const u32 own_l4 = lane4;
    lane4 = l4;
    PL<u32> in, node, ok, tnew, ign, ltt, dups, slot, sepoch;
    LBFT_FOR_LANES(l) {
      in[l] = l < avail ? 1u : 0u;
      node[l] = 0; ok[l] = 0; tnew[l] = 0xffffffffu; ign[l] = 0; ltt[l] = 0; dups[l] = 0; slot[l] = 0; sepoch[l] = 0;
      if (in[l]) {
        const u32 meta = ld(chw(c, pos + l));
        const u32 nd = meta & 0xffu;
        node[l] = nd;
        begin_node(nd);
        const i64 startup = (i64)(i32)nf(nd, NF_STARTUP);
        i64 next;
        ok[l] = update_is_noop(nd, (i64)clk - startup, next) ? 1u : 0u;
        if (RSPRUNQ) {  // the record exchange: the response's slot holds the peer's store -- and this event's is its only reference (coop_answer_requests, send_loop)
          const u32 sl = meta >> 16;
          const u32 ew = ld(sfw(sl, S_EPOCH));
          const Resp rp = load_response(sl);
          slot[l] = sl; sepoch[l] = ew;
          if ((ew >> 16) != 1u || !response_is_inert(nd, rp, sl)) ok[l] = 0;
        }
        i64 t_new = (i64)((u64)next + (u64)startup);  // process_node_actions (simulator.rs:296-325)
        if (t_new < (i64)clk + 1) t_new = (i64)clk + 1;
        i64 ig = t_new - 1;
        if (ig > (i64)P.max_clock) ig = P.max_clock;
        ign[l] = (u32)(i32)ig;
        tnew[l] = t_new <= (i64)P.max_clock ? (u32)t_new : 0xffffffffu;
        ltt[l] = nf(nd, NF_LAST_TIMER_T); dups[l] = nf(nd, NF_TIMER_DUPS);
      }
    }
    PL<u32> bad;
    LBFT_FOR_LANES(l) bad[l] = (in[l] && !ok[l]) ? 1u : 0u;
    const u64 BAD = pl_ballot(bad);
    const u32 cnt = BAD ? ctz64(BAD) : avail;
    LBFT_STAT(13); LBFT_STATN(14, cnt); if (BAD) LBFT_STAT(15); if (!cnt) LBFT_STAT(29);
    if (cnt) {
      const u64 RUN = (1ULL << cnt) - 1ULL;  // (cnt <= 31)
      PL<u64> MS;  // lanes of the run whose events are for this lane's node
      LBFT_FOR_LANES(l) MS[l] = l < cnt ? RUN : 0;
      for (u32 b = 0; (1u << b) < NN(); b++) {
        PL<u32> bit;
        LBFT_FOR_LANES(l) bit[l] = (l < cnt && ((node[l] >> b) & 1u)) ? 1u : 0u;
        const u64 B = pl_ballot(bit);
        if (B == 0 || B == RUN) continue;
        LBFT_FOR_LANES(l) MS[l] &= bit[l] ? B : ~B;
      }
      PL<u32> live, fold, tmeta;
      LBFT_FOR_LANES(l) {
        live[l] = 0; fold[l] = 0; tmeta[l] = node[l];  // a timer's event word: its node (no sender, no slot)
        if (l < cnt) {
          const bool first = (MS[l] & ((1ULL << l) - 1ULL)) == 0;
          const bool sched = tnew[l] != 0xffffffffu;               // (past the horizon: neither queued nor folded, the stamp is consumed)
          const bool first_folds = sched && tnew[l] == ltt[l];     // the node's pending timer has this very time
          live[l] = (first && sched && !first_folds) ? 1u : 0u;
          fold[l] = (sched && (!first || first_folds)) ? 1u : 0u;
          if (first) {
            const u32 g = popc64(MS[l]), last_l = 63u - (u32)clz64(MS[l]);
            const u32 folds = sched ? (first_folds ? g : g - 1u) : 0u;
            st(nfw(node[l], NF_IGNORE_UNTIL), ign[l]);
            if (sched && !first_folds) st(nfw(node[l], NF_LAST_TIMER_T), tnew[l]);
            if (folds) { st(nfw(node[l], NF_TIMER_DUPS), dups[l] + folds); st(nfw(node[l], NF_DUP_STAMP), stamp0 + last_l); }
          }
          if (RSPRUNQ) {  // snap_release: the slot's one reference goes, the slot returns to the stack -- in event order
            st(sfw(slot[l], S_EPOCH), sepoch[l] & 0xffffu);
            st(OFFSFREE() + sf0 + l, slot[l]);
          }
        }
      }
      const u64 F = pl_ballot(fold);
      if (is_k) qlen -= cnt;  // (the pops first: the queue's high-water mark is what the event-by-event order reaches)
      bulk_append(k, l4, clk, 0u /* 3 - UpdateTimer */, live, tnew, tmeta);
      lane4 = own_l4;
      if (is_k) {
        ev2 += cnt;
        if (RSPRUNQ) snap_free += cnt;
#if !defined(LBFT_NO_EXEC_COUNTERS)
        n_upd += cnt;
        n_fold += popc64(F);
#endif
        stamp += cnt;
        if (stamp >= (1u << 30)) fault |= F_STAMP_OVERFLOW;
        cal_cursor = idx;  // (bulk_append lowered the cursor to the timers' kind at this time; every timer lies at a later time)
        cal_advance(cnt);
      }
    }
    lane4 = own_l4;
    if (is_k) rsp_done = cnt;
    LBFT_FOR_LANES(l) if (l < cnt) LBFT_HOOK_POP(clk, 2u, node[l], 0u);  // (host analysis tools: the run's events are pops too)
    LBFT_MARK(5);  // (diagnostic builds: a response run is charged to the responses' phase)
  }
  // ---- cooperative notification run (round 6).  56-66 % of a large network's notifications leave their node exactly as it was -- every notification carries
  // the sender's certificates, proposal, vote and timeouts, and most receivers hold all of it already -- and they come in streaks (tests/tools/run_stats.cpp:
  // 70-85 % of them in streaks of >= 8 of one bucket).  notification_is_inert proves it per event from the snapshot's fixed words, the node's staged words and,
  // for the (at most four) blocks the snapshot names, the round / epoch words and the node's bit in the "known" / "QC" sets -- loads of immutable words and of
  // bits that only this node's events set; it is conservative wherever handle_notification would have to look further (a set extension word, a timeout's
  // highest_certified_block_round).  With update_is_noop the event then only releases its reference to the snapshot and schedules / folds the node's timer.
  // cert: the certificate block of a notification in the given roles (data_sync.rs:125-148)
  LBFT_HD bool cert_is_inert(u32 node, u32 b, bool as_hcc, bool as_hqc, u32 epoch) const {
    const u32 bb = boff(bfw(b, 0));
    const u32 r = ldf(bb, B_ROUND), qe = ldf(bb, B_EPOCH);
    u32 kw, qw;
    if (!wide() || node < 32) { kw = ldf(bb, B_KNOWN); qw = ldf(bb, B_QC); }
    else { kw = ld(bxw(b, B_KNOWN, node >> 5)); qw = ld(bxw(b, B_QC, node >> 5)); }
    const bool known = ((kw >> (node & 31u)) & 1u) != 0, hasqc = ((qw >> (node & 31u)) & 1u) != 0;
    if (qe > epoch) return false;          // should_sync
    if (qe < epoch) return true;           // not of this store: nothing inserted, nothing asked
    if (known && !hasqc) return false;     // insert_qc would insert it ("already inserted" / "the certified block must be verified first" return untouched)
    if (as_hcc && r > nf(node, NF_HC_ROUND) + 2) return false;
    if (as_hqc && r > nf(node, NF_HQC_ROUND)) return false;
    return true;
  }
  LBFT_HD bool notification_is_inert(u32 node, u32 sender, const Snap& sn) const {
    const u32 epoch = nf(node, NF_EPOCH), n_epoch = sn.w[S_EPOCH];
    if (n_epoch > epoch) return false;     // should_sync
    const u32 certs = sn.w[S_CERTS], hcc = certs & 0xffffu, hqc = certs >> 16;
    const bool c_hcc = hcc != 0 && hcc != nf(node, NF_HCC_BLK), c_hqc = hqc != 0 && hqc != nf(node, NF_HQC_BLK);
    bool ok = true;
    if (c_hcc) ok = cert_is_inert(node, hcc, true, c_hqc && hqc == hcc, epoch);
    if (ok && c_hqc && !(c_hcc && hqc == hcc)) ok = cert_is_inert(node, hqc, false, true, epoch);
    if (!ok) return false;
    if (n_epoch != epoch) return true;     // the rest of a notification is only looked at within one epoch (data_sync.rs:149)
    const u32 pv = sn.w[S_PROP_VOTE], pb = pv & 0xffffu, vote = pv >> 16;
    const u32 xk = node >> 5;
    if (pb) {  // insert_block returns at once for a block the node holds
      const u32 kw = (!wide() || node < 32) ? ld(bfw(pb, B_KNOWN)) : ld(bxw(pb, B_KNOWN, xk));
      if (!((kw >> (node & 31u)) & 1u)) return false;
    }
    const u32 cur = nf(node, NF_CUR_ROUND);
    const u32 tc_raw = sn.w[S_TC_ROUND], to_raw = sn.w[S_TO_ROUND];
    const u32 tc_round = wide() ? tc_raw & ~LBFT_S_XFLAG : tc_raw, to_round = wide() ? to_raw & ~LBFT_S_XFLAG : to_raw;
    const u32 have0 = nf(node, NF_TO_MASK);
    if (tc_round == cur && ((sn.w[S_TC_MASK] & ~have0) != 0 || (wide() && (tc_raw & LBFT_S_XFLAG) != 0))) return false;
    if (to_round == cur && ((sn.w[S_TO_MASK] & ~have0) != 0 || (wide() && (to_raw & LBFT_S_XFLAG) != 0))) return false;
    if (vote) {  // insert_vote: unknown block, another round, or an author already counted -> untouched
      const u32 vb = boff(bfw(vote, 0));
      const u32 vr = ldf(vb, B_ROUND);
      const u32 kw = (!wide() || node < 32) ? ldf(vb, B_KNOWN) : ld(bxw(vote, B_KNOWN, xk));
      const bool known = ((kw >> (node & 31u)) & 1u) != 0;
      if (known && vr == cur && !am_test(node, NF_BAL0_AUTHORS, sender) && !am_test(node, NF_BAL1_AUTHORS, sender)) return false;
    }
    return true;
  }
  // ---- notifications that change their node's current timeouts and / or ballot and nothing else (round 6, second form of the notification runs).  After the
  // runs above, nine in ten of the notifications that still took an ordinary step did exactly this (tests/tools/run_stats.cpp): the sender's own timeout and
  // its vote reach a receiver that holds everything else the notification carries -- insert_timeout (record_store.rs:390-415) adds an author to the current
  // timeouts without completing a certificate, insert_vote (:292-329) adds one to a ballot without winning the election, the update that follows is a no-op.
  // Such an event writes words of ITS node only (no block record, no message), so it can be a lane of a run like an inert one -- as long as it is the only
  // event of its node in the run (coop_notifications cuts the run in front of a second one).  notification_effects replays the tail of handle_notification
  // (data_sync.rs:149-169) on the node's STAGED words, in the reference's order (timeout-certificate set, current timeouts, vote; authors ascending), so that
  // update_is_noop then sees the node as the event leaves it; nothing reaches memory here.  What the lane has to write if it stays in the run is handed back
  // as a delta.  0 = the ordinary step must take the event (it does more, or more new timeouts than a delta carries), 1 = nothing changes, 2 = see `d`.
  struct NtfDelta {
    u32 acc[4];  // accepted timeouts: author | highest_certified_block_round << 8
    u32 flags;   // bits 0-2: accepted timeouts, bit 3: the node's NF_TC_SEL, bits 4-5: 1 + ballot entry the vote went to (0: no vote), bit 6: the election is ongoing (weights count)
    u32 to_w, bal_w;  // the new NF_TO_WEIGHT / NF_BALx_WEIGHT
  };
  LBFT_HD bool stage_timeouts(u32 node, u32 slot, u32 which, u32 w0, bool xf, NtfDelta& d) const {  // false: the ordinary step
    const u32 word0 = sfw(slot, S_FIXED_WORDS + which * NN());
    for (u32 k = 0; k < (xf ? MW() : 1u); k++) {
      u32 m = (k == 0 ? w0 : ld(sxw(slot, which, k))) & ~am_word(node, NF_TO_MASK, k);
      while (m) {
        const u32 a = 32u * k + ctz32(m);
        m &= m - 1;
        const u32 h = ld(word0 + a);
        if (h > nf(node, NF_HQC_ROUND)) continue;  // (insert_timeout's first test; the round is the current round, the author is new)
        const u32 n = d.flags & 7u;
        if (n >= 4u || h >= (1u << 24)) return false;
        const u32 v = a | (h << 8);
        d.acc[0] = n == 0 ? v : d.acc[0]; d.acc[1] = n == 1 ? v : d.acc[1]; d.acc[2] = n == 2 ? v : d.acc[2]; d.acc[3] = n == 3 ? v : d.acc[3];
        d.flags++;
        am_set(node, NF_TO_MASK, a);
        const u32 w = nf(node, NF_TO_WEIGHT) + weight(node, a);
        nfs(node, NF_TO_WEIGHT, w);
        if (w >= QUORUM()) return false;  // a timeout certificate forms: the round moves
      }
    }
    return true;
  }
  LBFT_HD u32 notification_effects(u32 node, u32 sender, u32 slot, const Snap& sn, NtfDelta& d) const {
    d.acc[0] = d.acc[1] = d.acc[2] = d.acc[3] = 0; d.flags = 0; d.to_w = 0; d.bal_w = 0;
    const u32 epoch = nf(node, NF_EPOCH), n_epoch = sn.w[S_EPOCH];
    if (n_epoch > epoch) return 0;         // should_sync
    const u32 certs = sn.w[S_CERTS], hcc = certs & 0xffffu, hqc = certs >> 16;
    const bool c_hcc = hcc != 0 && hcc != nf(node, NF_HCC_BLK), c_hqc = hqc != 0 && hqc != nf(node, NF_HQC_BLK);
    bool ok = true;
    if (c_hcc) ok = cert_is_inert(node, hcc, true, c_hqc && hqc == hcc, epoch);
    if (ok && c_hqc && !(c_hcc && hqc == hcc)) ok = cert_is_inert(node, hqc, false, true, epoch);
    if (!ok) return 0;
    if (n_epoch != epoch) return 1;        // the rest of a notification is only looked at within one epoch (data_sync.rs:149)
    const u32 pv = sn.w[S_PROP_VOTE], pb = pv & 0xffffu, vote = pv >> 16;
    const u32 xk = node >> 5;
    if (pb) {  // insert_block returns at once for a block the node holds
      const u32 kw = (!wide() || node < 32) ? ld(bfw(pb, B_KNOWN)) : ld(bxw(pb, B_KNOWN, xk));
      if (!((kw >> (node & 31u)) & 1u)) return 0;
    }
    const u32 cur = nf(node, NF_CUR_ROUND);
    const u32 tc_raw = sn.w[S_TC_ROUND], to_raw = sn.w[S_TO_ROUND];
    const u32 tc_round = wide() ? tc_raw & ~LBFT_S_XFLAG : tc_raw, to_round = wide() ? to_raw & ~LBFT_S_XFLAG : to_raw;
    if (tc_round == cur && !stage_timeouts(node, slot, 0, sn.w[S_TC_MASK], wide() && (tc_raw & LBFT_S_XFLAG) != 0, d)) return 0;
    if (to_round == cur && !stage_timeouts(node, slot, 1, sn.w[S_TO_MASK], wide() && (to_raw & LBFT_S_XFLAG) != 0, d)) return 0;
    if (vote) {  // insert_vote: unknown block, another round, or an author already counted -> untouched
      const u32 vb = boff(bfw(vote, 0));
      const u32 vr = ldf(vb, B_ROUND);
      const u32 kw = (!wide() || node < 32) ? ldf(vb, B_KNOWN) : ld(bxw(vote, B_KNOWN, xk));
      const bool known = ((kw >> (node & 31u)) & 1u) != 0;
      if (known && vr == cur && !am_test(node, NF_BAL0_AUTHORS, sender) && !am_test(node, NF_BAL1_AUTHORS, sender)) {
        const u32 b0 = nf(node, NF_BAL0_BLK), b1 = nf(node, NF_BAL1_BLK);
        const bool ongoing = (nf(node, NF_ELECTION) & 0xffu) == 0;
        u32 w = 0;
        if (b0 == vote || b0 == 0) {
          nfs(node, NF_BAL0_BLK, vote); am_set(node, NF_BAL0_AUTHORS, sender);
          if (ongoing) { w = nf(node, NF_BAL0_WEIGHT) + weight(node, sender); nfs(node, NF_BAL0_WEIGHT, w); }
          d.flags |= 1u << 4;
        } else if (b1 == vote || b1 == 0) {
          nfs(node, NF_BAL1_BLK, vote); am_set(node, NF_BAL1_AUTHORS, sender);
          if (ongoing) { w = nf(node, NF_BAL1_WEIGHT) + weight(node, sender); nfs(node, NF_BAL1_WEIGHT, w); }
          d.flags |= 2u << 4;
        } else return 0;                   // (a third ballot entry: the ordinary step reports the overflow)
        if (ongoing && w >= QUORUM()) return 0;  // the election is won: check_for_new_qc has work
        d.flags |= ongoing ? 1u << 6 : 0u;
        d.bal_w = w;
      }
    }
    if (!(d.flags & 0x37u)) return 1;
    d.flags |= (nf(node, NF_TC_SEL) & 1u) << 3;
    d.to_w = nf(node, NF_TO_WEIGHT);
    return 2;
  }
  // ... and the delta written to the node's rows (what end_node would write of the staged words: the set words by read-modify-write)
  LBFT_HD void commit_notification_delta(u32 node, u32 sender, u32 vote, u32 a0, u32 a1, u32 a2, u32 a3, u32 flags, u32 to_w, u32 bal_w) const {
    const u32 n = flags & 7u, buf = 1u - ((flags >> 3) & 1u);
    LBFT_UNROLL
    for (u32 j = 0; j < 4; j++) {
      if (j < n) {
        const u32 v = j == 0 ? a0 : j == 1 ? a1 : j == 2 ? a2 : a3;
        const u32 a = v & 0xffu, h = v >> 8;
        const u32 addr = (!wide() || a < 32) ? nfw(node, NF_TO_MASK) : amxw(node, NF_TO_MASK, a >> 5);
        st(addr, ld(addr) | (1u << (a & 31u)));
        hc_set(node, buf, a, h);
      }
    }
    if (n) st(nfw(node, NF_TO_WEIGHT), to_w);
    const u32 e = (flags >> 4) & 3u;
    if (e) {
      const bool e0 = e == 1u;
      st(nfw(node, e0 ? NF_BAL0_BLK : NF_BAL1_BLK), vote);
      const u32 f = e0 ? NF_BAL0_AUTHORS : NF_BAL1_AUTHORS;
      const u32 addr = (!wide() || sender < 32) ? nfw(node, f) : (e0 ? amxw(node, NF_BAL0_AUTHORS, sender >> 5) : amxw(node, NF_BAL1_AUTHORS, sender >> 5));
      st(addr, ld(addr) | (1u << (sender & 31u)));
      if (flags & (1u << 6)) st(nfw(node, e0 ? NF_BAL0_WEIGHT : NF_BAL1_WEIGHT), bal_w);
    }
  }
  u32 ntf_done;  // leader lane: notifications the last coop_notifications consumed
  u32 ntf_skip;  // leader lane: notifications behind it that were seen NOT to be inert -- they take ordinary steps without another attempt
  // ALL networks of the wavefront at once: the wavefront's 64 lanes are split into one SEGMENT of W = 64 / lpw lanes per network (network s = lane s, its
  // segment = lanes [s W, (s + 1) W)); lane j of a segment takes the j-th event at the head of its network's open bucket and addresses that network's column.
  // What a run of ONE network needs from its leader lane (head and tail words, clock, stamp, free-slot count) comes by shuffle, ballots are cut down to the
  // segment; every leader then books its own segment's result.  (One network after the other, the first form of this run, a run cost as much as an ordinary
  // step of the whole wavefront: c5 -4 %, c4 +7 % -- profiles/r06.)  A run never appends to the calendar: an event whose timer is NOT folded into the node's
  // pending one (1-2 % of them) ends the run like an event that is not inert and takes the ordinary step.  `want`: this lane's network has a notification
  // run to try; `budget`: events it may still process in this launch.  The host build (one network per object) is one segment of 64 lanes.
  LBFT_HD void coop_notifications(bool want, u32 budget) {
#if defined(__HIP_DEVICE_COMPILE__)
    const u32 lane_ = lbft_lane_id();
    const u32 W = 64u / P.lpw;                       // (lpw is a power of two <= 32 here: run_coop asks for W >= 2)
    const u32 seg = lane_ / W, seg_sh = seg * W;     // the network this lane works for, first lane of its segment
    const u32 lead_sh = lane_ * W;                   // (leader role) first lane of the segment that works for THIS lane's network
    const bool leader = lane_ < P.lpw;
#define LBFT_SEGV(x) ((u32)__shfl((int)(x), (int)seg, 64))
#else
    const u32 W = 64u, seg_sh = 0u, lead_sh = 0u;
    const bool leader = true;
#define LBFT_SEGV(x) ((u32)(x))
#endif
    const u64 wmask = W >= 64u ? ~0ULL : (1ULL << (W & 63u)) - 1ULL;
    // leader side: what its segment may take
    u32 own_avail = 0;
    if (leader && want) {
      own_avail = ((sp_nx >> 6) == (cur_h >> 6) ? (sp_nx & 63u) : LBFT_CAL_CE) - (cur_h & 63u) + 1u;
      if (own_avail > budget) own_avail = budget;
      if (own_avail > W) own_avail = W;
      if ((i32)(sp_idx >> 2) > clock) clock = (i32)(sp_idx >> 2);
    }
    // worker side
    const u32 avail = LBFT_SEGV(own_avail), l4 = LBFT_SEGV(lane4), h = LBFT_SEGV(cur_h);
    const i32 clk = (i32)LBFT_SEGV((u32)clock);
    const u32 stamp0 = LBFT_SEGV(stamp), sf0 = LBFT_SEGV(snap_free);
    const u32 c = (h >> 6) - 1u, pos = h & 63u;
    const u32 own_l4 = lane4;
    lane4 = l4;
    PL<u32> in, node, slot, ok, tnew, ign, dups, refs, sepoch;
    PL<u32> act, da0, da1, da2, da3, dfl, dtw, dbw;  // (NTFACT) the lane's event changes its node's timeouts / ballot: the delta it writes if it stays in the run
    LBFT_FOR_LANES(l) {
      const u32 j = l - seg_sh;
      in[l] = j < avail ? 1u : 0u;
      node[l] = 0; slot[l] = 0; ok[l] = 0; tnew[l] = 0xffffffffu; ign[l] = 0; dups[l] = 0; refs[l] = 0; sepoch[l] = 0;
      act[l] = 0; da0[l] = 0; da1[l] = 0; da2[l] = 0; da3[l] = 0; dfl[l] = 0; dtw[l] = 0; dbw[l] = 0;
      if (in[l]) {
        const u32 meta = ld(chw(c, pos + j));
        const u32 nd = meta & 0xffu, sender = (meta >> 8) & 0xffu, sl = meta >> 16;
        node[l] = nd; slot[l] = sl;
        begin_node(nd);
        const Snap sn = load_snapshot(sl);
        refs[l] = sn.refs; sepoch[l] = sn.w[S_EPOCH];
        u32 cls;  // 0: the ordinary step, 1: nothing of the node changes, 2: its timeouts / ballot do (staged: update_is_noop sees the node as the event leaves it)
        if (NTFACT) {
          NtfDelta d;
          cls = notification_effects(nd, sender, sl, sn, d);
          if (cls == 2u) { act[l] = 1u; da0[l] = d.acc[0]; da1[l] = d.acc[1]; da2[l] = d.acc[2]; da3[l] = d.acc[3]; dfl[l] = d.flags | ((sn.w[S_PROP_VOTE] >> 16) << 16) | (sender << 8); dtw[l] = d.to_w; dbw[l] = d.bal_w; }
        } else cls = notification_is_inert(nd, sender, sn) ? 1u : 0u;
        const i64 startup = (i64)(i32)nf(nd, NF_STARTUP);
        i64 next;
        const bool quiet = update_is_noop(nd, (i64)clk - startup, next);
        i64 t_new = (i64)((u64)next + (u64)startup);  // process_node_actions (simulator.rs:296-325)
        if (t_new < (i64)clk + 1) t_new = (i64)clk + 1;
        i64 ig = t_new - 1;
        if (ig > (i64)P.max_clock) ig = P.max_clock;
        ign[l] = (u32)(i32)ig;
        tnew[l] = t_new <= (i64)P.max_clock ? (u32)t_new : 0xffffffffu;
        dups[l] = nf(nd, NF_TIMER_DUPS);
        const bool folds = tnew[l] == 0xffffffffu || tnew[l] == nf(nd, NF_LAST_TIMER_T);  // (past the horizon: nothing is queued either)
        ok[l] = (quiet && folds && cls != 0u) ? 1u : 0u;
      }
    }
    PL<u32> bad;
    LBFT_FOR_LANES(l) bad[l] = (in[l] && !ok[l]) ? 1u : 0u;
    const u64 BAD = pl_ballot(bad);
    const u64 segbad = (BAD >> seg_sh) & wmask;
    u32 cnt = segbad ? ctz64(segbad) : avail;
    u64 RUN = cnt ? (((cnt >= 64u ? ~0ULL : (1ULL << cnt) - 1ULL)) << seg_sh) : 0ULL;  // the lanes of this segment's run
    PL<u64> MS, SS;  // lanes of the run whose events are for this lane's node / hold a reference to this lane's snapshot slot
    LBFT_FOR_LANES(l) MS[l] = ((RUN >> l) & 1ULL) ? RUN : 0;
    for (u32 b = 0; (1u << b) < NN(); b++) {
      PL<u32> bit;
      LBFT_FOR_LANES(l) bit[l] = (MS[l] != 0 && ((node[l] >> b) & 1u)) ? 1u : 0u;
      const u64 B = pl_ballot(bit);
      LBFT_FOR_LANES(l) MS[l] &= bit[l] ? B : ~B;
    }
    u64 CONF = 0;
    if (NTFACT) {
      // an event that changes its node must be the only event of that node in the run (the others' predicates and timers were evaluated on the node
      // as it was): the run ends in front of a later event of a node that an event of the run changes, and in front of a changing event that is not
      // its node's first -- that one heads the next run
      const u64 ACT = pl_ballot(act);
      PL<u32> conf;
      LBFT_FOR_LANES(l) {
        const u64 earlier = MS[l] & ((1ULL << l) - 1ULL);
        conf[l] = (earlier != 0 && (act[l] || (earlier & ACT) != 0)) ? 1u : 0u;
      }
      CONF = pl_ballot(conf);
      const u64 segconf = (CONF >> seg_sh) & wmask;
      if (segconf) {
        cnt = ctz64(segconf);  // (>= 1: an event has no earlier one to collide with unless the run has two)
        RUN = ((1ULL << cnt) - 1ULL) << seg_sh;
        LBFT_FOR_LANES(l) MS[l] = ((RUN >> l) & 1ULL) ? (MS[l] & RUN) : 0;
      }
    }
    LBFT_STAT(60); LBFT_STATN(61, cnt); if (!cnt) LBFT_STAT(62);
    LBFT_FOR_LANES(l) SS[l] = ((RUN >> l) & 1ULL) ? RUN : 0;
    for (u32 b = 0; (1u << b) < P.scap; b++) {
      PL<u32> bit;
      LBFT_FOR_LANES(l) bit[l] = (SS[l] != 0 && ((slot[l] >> b) & 1u)) ? 1u : 0u;
      const u64 B = pl_ballot(bit);
      LBFT_FOR_LANES(l) SS[l] &= bit[l] ? B : ~B;
    }
    // the snapshots' references: the LAST event of a slot's group leaves the count behind (and frees the slot at zero -- in event order)
    PL<u32> freed, fold;
    LBFT_FOR_LANES(l) {
      freed[l] = 0; fold[l] = 0;
      if (MS[l] != 0) {
        if ((SS[l] >> l) == 1ULL) {  // no later lane of the run holds this slot
          const u32 left = refs[l] - popc64(SS[l]);
          st(sfw(slot[l], S_EPOCH), sepoch[l] | (left << 16));
          freed[l] = left == 0 ? 1u : 0u;
        }
        fold[l] = tnew[l] != 0xffffffffu ? 1u : 0u;
        if (NTFACT && act[l]) commit_notification_delta(node[l], (dfl[l] >> 8) & 0xffu, dfl[l] >> 16, da0[l], da1[l], da2[l], da3[l], dfl[l] & 0xffu, dtw[l], dbw[l]);
        if ((MS[l] & ((1ULL << l) - 1ULL)) == 0) {  // first event of its node in the run: the node's timer words, once
          st(nfw(node[l], NF_IGNORE_UNTIL), ign[l]);
          if (fold[l]) {
            const u32 last_l = 63u - (u32)clz64(MS[l]);
            st(nfw(node[l], NF_TIMER_DUPS), dups[l] + popc64(MS[l]));
            st(nfw(node[l], NF_DUP_STAMP), stamp0 + (last_l - seg_sh));
          }
        }
      }
    }
    const u64 FR = pl_ballot(freed), F = pl_ballot(fold);
    LBFT_FOR_LANES(l) if (freed[l]) st(OFFSFREE() + sf0 + popc64(FR & (wmask << seg_sh) & ((1ULL << l) - 1ULL)), slot[l]);
    lane4 = own_l4;
    // leader side: book the segment's result
    if (leader && want) {
      const u64 mybad = (BAD >> lead_sh) & wmask;
      u32 mycnt = mybad ? ctz64(mybad) : own_avail;
      ntf_skip = mybad ? ctz64(~(mybad >> mycnt)) : 0u;  // (consecutive events from the cut on that were seen to need an ordinary step)
      if (NTFACT) {
        const u64 myconf = (CONF >> lead_sh) & wmask;
        if (myconf) { mycnt = ctz64(myconf); ntf_skip = 0; }  // (cut in front of a collision: that event heads the next run)
      }
      ntf_done = mycnt;
      if (mycnt) {
        snap_free += popc64((FR >> lead_sh) & wmask);
        qlen -= mycnt;
        ev0 += mycnt;
#if !defined(LBFT_NO_EXEC_COUNTERS)
        n_upd += mycnt;
        n_fold += popc64((F >> lead_sh) & wmask);
#endif
        stamp += mycnt;
        if (stamp >= (1u << 30)) fault |= F_STAMP_OVERFLOW;
        cal_advance(mycnt);
      }
    }
    LBFT_FOR_LANES(l) if (MS[l] != 0) LBFT_HOOK_POP(clk, 0u, node[l], 0u);
#undef LBFT_SEGV
    LBFT_MARK(3);  // (diagnostic builds: a notification run is charged to the notifications' first phase)
  }
  u32 bulk;       // leader lane: bit 0 = a broadcast is pending, bit 1 = a query-all is pending (set by send_loop)
  u32 bulk_copy;  // leader lane: bit 0 = the hcbr words of a response snapshot are to be copied (a request under quirks bit 0), slot << 8
  u32 bulk_node;  // leader lane: the node whose actions are being processed
  LBFT_HD u32 ldc(u32 l4, u32 w) const { LBFT_HOOK_MEM((IMAJOR ? (w << 2) : mul24(w, rowb())) + l4, 0); return *reinterpret_cast<const u32*>(tile + (size_t)((IMAJOR ? (w << 2) : mul24(w, rowb())) + l4)); }
  LBFT_HD void stc(u32 l4, u32 w, u32 v) const { LBFT_HOOK_MEM((IMAJOR ? (w << 2) : mul24(w, rowb())) + l4, 1); *reinterpret_cast<u32*>(tile + (size_t)((IMAJOR ? (w << 2) : mul24(w, rowb())) + l4)) = v; }
  // the first try of sample_delay() on the draw `bits`: true = accepted (then d is the delay sample_delay() returns)
  LBFT_HD bool fast_delay(u64 bits, i64& d) const {
    if (DMODEL() == 1) {
      u64 zone = (P.uni_span << clz64(P.uni_span)) - 1;
      d = P.uni_lo + (i64)mulhi64(bits, P.uni_span);
      return bits * P.uni_span <= zone;
    }
    u32 i = (u32)(bits & 0xff);
    double u = lbft_asdouble((1024ULL << 52) | (bits >> 12)) - 3.0;
    double x = u * lbft_asdouble(zig_x[i]);
    double ax = x < 0.0 ? -x : x;
    d = trunc_exp(P.mu + P.sigma * x);
    return ax < lbft_asdouble(zig_x[i + 1]);
  }
  LBFT_HD u32 horizon_time(i32 clk, i64 d) const {  // clock + delay, 0xffffffff = past max_clock (never queued)
    i64 t = d > INT64_MAX - (i64)clk ? INT64_MAX : (i64)clk + d;
    return t <= (i64)P.max_clock ? (u32)t : 0xffffffffu;
  }
  // ---- the receivers' shuffle of a bulk send, lanes = STEPS (round 6, third session).  SliceRandom::shuffle is `for i = cnt - 1 .. 1: swap(i, j_i)` with
  // j_i = gen_range(0, i + 1) off the RNG (simulator.rs:343,370): a chain of ~1.2 cnt dependent draws and cnt dependent swaps -- on the scalar unit 50
  // instructions per draw, a quarter of a 100-node network's time after the runs.  Both halves have a lane-parallel form that yields the SAME list:
  //   (1) which draw serves which step.  Draw d is accepted for step x iff low32(v_d (x + 1)) <= zone(x + 1), and the step a draw is tried for is
  //       i0 - (accepted draws before it): a fixed point over the ballot of accepted lanes, exact because lane l's step depends on lanes < l only (one more
  //       lane of the prefix is right after every iteration; it stops at once when nothing was rejected).  The accepted draws are then moved from draw order
  //       to step order (step s takes the (i0 - s)-th accepted draw: a select on the ballot by binary search).
  //   (2) the list from the j_i.  Position i is final after step i and holds what position j_i held just before: follow that content back -- position p at
  //       time t was last written by the nearest later-executed... EARLIER-executed step s > t with j_s == p, which put there what position s held before step s, and so on:
  //       final[i] = init[e_i], e_i = the end of the chain  s1 = min{s > i : j_s == j_i},  s_{k+1} = up(s_k) = min{s > s_k : j_s == s_k}  (e_i = j_i without an s1).
  //       Both relations are equality matches between lanes: bit-sliced ballots over the 7 bits of j; the chains are closed by pointer jumping.
  // Steps live in two registers (step l in the first, step 64 + l in the second); position 0 counts as a step that swaps with itself.
  LBFT_HD static u64 bits_from(u32 b) { return b >= 64u ? 0ULL : (~0ULL << b); }
  LBFT_HD u32 step_above(u64 lo, u64 hi, u32 i) const {  // the lowest set bit above step i of the 128 steps (lo, hi); 0xff: none
    const u64 a = i < 64u ? lo & bits_from(i + 1u) : 0ULL;
    const u64 b = i < 64u ? hi : hi & bits_from(i - 63u);
    return a ? ctz64(a) : b ? 64u + ctz64(b) : 0xffu;
  }
  LBFT_HD void coop_shuffle(u32 k, u32 l4, u32 node, u32 cnt, PL<u32>& perm0, PL<u32>& perm1) {
    const bool is_k = LBFT_IS_LANE(k);
    const u32 ring_row = P.off_ring, ring_mask = P.ring - 1u;
    const bool two = cnt > 64u;
    PL<u32> J0, J1;  // j of step l / step 64 + l
    LBFT_FOR_LANES(l) { J0[l] = l; J1[l] = 64u + l; }
    // (1) the draws
    for (u32 i0 = cnt - 1u; i0 >= 1u;) {
      u32 avail = LBFT_UNI(rng.rcnt, k);
      if (avail == 0) {
        if (is_k) rng.ring_fill(rng.ring_room() < 64u ? rng.ring_room() : 64u);
        avail = LBFT_UNI(rng.rcnt, k);
      }
      const u32 take = avail < 64u ? avail : 64u;
      const u32 head = LBFT_UNI(rng.rhead, k);
      PL<u32> dhi;  // next_u32() = next_u64() >> 32 of the next `take` draws
      LBFT_FOR_LANES(l) dhi[l] = l < take ? ldc(l4, ring_row + 2u * ((head + l) & ring_mask) + 1u) : 0u;
      u64 AM = take >= 64u ? ~0ULL : ((1ULL << take) - 1ULL);  // first guess: every draw is accepted
      PL<u32> jv;
      for (;;) {
        PL<u32> acc;
        LBFT_FOR_LANES(l) {
          const u32 before = popc64(AM & ((1ULL << l) - 1ULL));
          const bool valid = l < take && before < i0;  // (the shuffle ends with step 1: later draws are not consumed)
          const u32 range = valid ? i0 - before + 1u : 2u, zone = (range << clz32(range)) - 1u;
          const u64 mm = (u64)dhi[l] * range;
          acc[l] = (valid && (u32)mm <= zone) ? 1u : 0u;
          jv[l] = (u32)(mm >> 32);
        }
        const u64 NM = pl_ballot(acc);
        if (NM == AM) break;
        AM = NM;
      }
      const u32 na = popc64(AM);
      const u32 used = na == i0 ? 64u - (u32)clz64(AM) : take;  // (all steps served: the draws behind the last accepted one stay in the ring)
      // draw order -> step order: step s = i0 - r takes the r-th accepted draw
      for (u32 h = 0; h < (two ? 2u : 1u); h++) {
        PL<u32> src, got;
        LBFT_FOR_LANES(l) {
          const u32 s = h * 64u + l;
          u32 r = i0 - s, pos = 0;  // (wraps for s > i0: not < na)
          if (s <= i0 && r < na) {
            LBFT_UNROLL
            for (u32 b = 32u; b >= 1u; b >>= 1) {
              const u32 c = popc64(AM & (((1ULL << b) - 1ULL) << pos));
              if (r >= c) { r -= c; pos += b; }
            }
          }
          src[l] = pos;
        }
        pl_shfl(got, jv, src);
        LBFT_FOR_LANES(l) {
          const u32 s = h * 64u + l, r = i0 - s;
          if (s <= i0 && r < na) { if (h == 0) J0[l] = got[l]; else J1[l] = got[l]; }
        }
      }
      if (is_k) { rng.rhead += used; rng.rcnt -= used; rng.draws += used; }
      i0 -= na;
    }
    // (2) the list
    const u64 V0 = cnt >= 64u ? ~0ULL : ((1ULL << cnt) - 1ULL), V1 = two ? ((1ULL << (cnt - 64u)) - 1ULL) : 0ULL;  // the steps 0 .. cnt - 1
    // (all eight masks at once: taken one relation and one register of steps at a time -- two masks live -- the kernels came out with MORE spilled registers
    // and 1-2 % slower: profiles/r06/parallel_shuffle_ab.txt)
    PL<u64> S0l, S0h, U0l, U0h, S1l, S1h, U1l, U1h;  // register 0 / 1 lanes: steps with the same j (S) / with j == this step (U), low and high 64 steps
    LBFT_FOR_LANES(l) { S0l[l] = V0; S0h[l] = V1; U0l[l] = V0; U0h[l] = V1; S1l[l] = V0; S1h[l] = V1; U1l[l] = V0; U1h[l] = V1; }
    for (u32 b = 0; b < 7u; b++) {
      PL<u32> b0, b1;
      LBFT_FOR_LANES(l) { b0[l] = (J0[l] >> b) & 1u; b1[l] = two ? (J1[l] >> b) & 1u : 0u; }
      const u64 B0 = pl_ballot(b0), B1 = two ? pl_ballot(b1) : 0ULL;
      LBFT_FOR_LANES(l) {
        const bool i0b = ((l >> b) & 1u) != 0, i1b = (((64u + l) >> b) & 1u) != 0;
        S0l[l] &= b0[l] ? B0 : ~B0; U0l[l] &= i0b ? B0 : ~B0;
        if (two) {
          S0h[l] &= b0[l] ? B1 : ~B1; U0h[l] &= i0b ? B1 : ~B1;
          S1l[l] &= b1[l] ? B0 : ~B0; S1h[l] &= b1[l] ? B1 : ~B1;
          U1l[l] &= i1b ? B0 : ~B0; U1h[l] &= i1b ? B1 : ~B1;
        }
      }
    }
    PL<u32> s1a, s1b, p0, p1;  // the first link of a step's chain (0xff: none); up(step), a step without one pointing at itself
    LBFT_FOR_LANES(l) {
      s1a[l] = step_above(S0l[l], S0h[l], l);
      const u32 u = step_above(U0l[l], U0h[l], l);
      p0[l] = u == 0xffu ? l : u;
      s1b[l] = 0xffu; p1[l] = 64u + l;
      if (two) {
        s1b[l] = step_above(S1l[l], S1h[l], 64u + l);
        const u32 u1 = step_above(U1l[l], U1h[l], 64u + l);
        p1[l] = u1 == 0xffu ? 64u + l : u1;
      }
    }
    for (u32 rounds = 0; (1u << rounds) < cnt; rounds++) {  // pointer jumping: p = the end of the step's up-chain
      PL<u32> g00, g01, g10, g11;
      pl_shfl(g00, p0, p0);
      if (two) { pl_shfl(g01, p1, p0); pl_shfl(g10, p0, p1); pl_shfl(g11, p1, p1); }
      LBFT_FOR_LANES(l) {
        const u32 a = p0[l], b = p1[l];
        p0[l] = (two && a >= 64u) ? g01[l] : g00[l];
        if (two) p1[l] = b >= 64u ? g11[l] : g10[l];
      }
    }
    {
      PL<u32> ia, ib, e00, e01, e10, e11;
      LBFT_FOR_LANES(l) { ia[l] = s1a[l] == 0xffu ? 0u : s1a[l]; ib[l] = s1b[l] == 0xffu ? 0u : s1b[l]; }
      pl_shfl(e00, p0, ia);
      if (two) { pl_shfl(e01, p1, ia); pl_shfl(e10, p0, ib); pl_shfl(e11, p1, ib); }
      LBFT_FOR_LANES(l) {
        const u32 ea = s1a[l] == 0xffu ? J0[l] : ((two && s1a[l] >= 64u) ? e01[l] : e00[l]);
        perm0[l] = ea < node ? ea : ea + 1u;  // init[e]: the receivers in index order skip the sender
        u32 eb = 64u + l;
        if (two) eb = s1b[l] == 0xffu ? J1[l] : (s1b[l] >= 64u ? e11[l] : e10[l]);
        perm1[l] = eb < node ? eb : eb + 1u;
      }
    }
  }
  LBFT_HD void coop_bulk(u32 k, u32 which) {
    const bool is_k = LBFT_IS_LANE(k);
    const u32 l4 = LBFT_UNI(lane4, k);
    const u32 node = LBFT_UNI(bulk_node, k);
    const i32 clk = (i32)LBFT_UNI((u32)clock, k);
    const u32 cnt = NN() - 1;
    const u32 kc = which ? 2u : 3u;  // 3 - Event kind (DataSyncNotify = 0, DataSyncRequest = 1): the bucket within a time
    // leader: what the scalar loop decides at the start of a list
    u32 eq_k = 0, rs_k = 0;
    if (is_k) {
      if (which == 0) {
        if (is_equivocator(node)) {  // (E2): does this notification carry one of the node's own (double) proposals?
          u32 pb = proposed_block(node);
          eq_k = (pb != 0 && blk_get(pb).author() == node) ? 1u : 0u;
        }
      } else {
        rs_k = q1() ? (u32)make_request_slot(nf(node, NF_EPOCH), nf(node, NF_HCC_BLK) | (nf(node, NF_HQC_BLK) << 16)) : 0u;
      }
    }
    const u32 equivocal = LBFT_UNI(eq_k, k);
    const i32 rs = (i32)LBFT_UNI(rs_k, k);
    LBFT_CMARK(6);  // leader's prework
    // ---- receivers in index order, then SliceRandom::shuffle: for i = cnt - 1 .. 1: swap(i, gen_range_u32(i + 1)) ----
    PL<u32> perm0, perm1;  // entry i of the list: lane i of perm0 (i < 64) / lane i - 64 of perm1
    coop_shuffle(k, l4, node, cnt, perm0, perm1);
    LBFT_MARK(16);
    // ---- one delay sample per receiver, in list order ----
    PL<u32> tm0, tm1;  // scheduled time of message j (lane j of tm0 / lane j - 64 of tm1); 0xffffffff = past the horizon
    coop_sample(k, l4, clk, cnt, tm0, tm1, false);
    LBFT_MARK(18);
    // ---- schedule: lanes = messages, 64 at a time ----
    i32 sr = -1, st_ = -1;  // snapshot slot of the notification / of its twin (-1 not needed yet, -2 none available)
    u32 refs = 0, refs_twin = 0, rrefs = 0;
    for (u32 jb = 0; jb < cnt; jb += 64u) {
      PL<u32> r, t, live, twin;
      LBFT_FOR_LANES(l) {
        r[l] = jb ? perm1[l] : perm0[l];
        t[l] = jb + l < cnt ? (jb ? tm1[l] : tm0[l]) : 0xffffffffu;
        live[l] = t[l] != 0xffffffffu ? 1u : 0u;
        twin[l] = (which == 0 && equivocal && (r[l] & 1u) == 0) ? 1u : 0u;
      }
      const u64 T = pl_ballot(twin);
      if (which == 0) {  // the notification snapshot(s), created at the first message that is actually scheduled
        const u64 L0 = pl_ballot(live);
        const u64 Lr = L0 & ~T, Lt = L0 & T;
        const bool need_r = Lr != 0 && sr == -1, need_t = Lt != 0 && st_ == -1;
        const bool twin_first = need_r && need_t && ctz64(Lt) < ctz64(Lr);
        if (is_k) {
          for (u32 q = 0; q < 2; q++) {
            bool do_twin = (q == 0) == twin_first;
            if (do_twin && need_t) { st_ = snap_alloc(); if (st_ < 0) st_ = -2; else write_snapshot(node, (u32)st_, true, true); }
            if (!do_twin && need_r) { sr = snap_alloc(); if (sr < 0) sr = -2; else write_snapshot(node, (u32)sr, false, true); }
          }
        }
        sr = (i32)LBFT_UNI((u32)sr, k); st_ = (i32)LBFT_UNI((u32)st_, k);
        if (need_t && st_ >= 0) coop_copy_hcbr(k, l4, node, (u32)st_);
        if (need_r && sr >= 0) coop_copy_hcbr(k, l4, node, (u32)sr);
        LBFT_FOR_LANES(l) if (live[l] && (twin[l] ? st_ < 0 : sr < 0)) live[l] = 0;  // no slot: not scheduled, the stamp is consumed
        LBFT_CMARK(7);  // snapshot(s)
      } else if (rs < 0) {
        LBFT_FOR_LANES(l) live[l] = 0;
      }
      PL<u32> meta;
      LBFT_FOR_LANES(l) {
        u32 snap = which == 0 ? (u32)(twin[l] ? st_ : sr) : (u32)rs;
        meta[l] = which == 0 ? (r[l] | (node << 8) | (snap << 16)) : (node | (r[l] << 8) | (snap << 16));
      }
      const u64 L = bulk_append(k, l4, clk, kc, live, t, meta);
      const u32 nl = popc64(L);
      if (nl) {
        if (which == 0) { refs += popc64(L & ~T); refs_twin += popc64(L & T); } else rrefs += nl;
      }
    }
    if (is_k) {
      stamp += cnt;
      if (stamp >= (1u << 30)) fault |= F_STAMP_OVERFLOW;
      if (which == 0) {
        if (sr >= 0) { if (refs) snap_set_refs((u32)sr, refs, nf(node, NF_EPOCH)); else snap_free_slot((u32)sr); }
        if (st_ >= 0) { if (refs_twin) snap_set_refs((u32)st_, refs_twin, nf(node, NF_EPOCH)); else snap_free_slot((u32)st_); }
      } else if (q1() && rs >= 0) {
        if (rrefs) snap_set_refs((u32)rs, rrefs, nf(node, NF_EPOCH)); else snap_free_slot((u32)rs);
      }
    }
    LBFT_MARK(19);
  }

  // ---- record hashes of a node's committed chain (first half: byte-exact record hashing) ----
  // For the k-th committed command of `node`: out[4k] = hash of the Block that carried it, [4k+1] = the State after it,
  // [4k+2] = hash of the QuorumCertificate certifying the block (votes in author order, as the oracle canonicalises the
  // reference's HashMap order), [4k+3] = number of votes | flags << 32 (bit 0: no QC recorded, bit 1: the block's
  // predecessor is not the previous commit -- cannot happen, commits extend one another).  Everything is recomputed from the
  // block pool: the structural ids the event loop works with never carried these hashes.  Returns the number of commits.
  LBFT_HD u32 committed_record_hashes(u32 node, u64* out, u32 cap) const {
    u32 nc = nfm(node, NF_NCOMMITS);
    u64 qc_prev = 0, state_prev = 0, state_prev2 = 0;
    u32 y_prev = 0, y_prev2 = 0;
    for (u32 k = 0; k < nc && k < cap; k++) {
      u32 y = ld(P.off_log + node * P.lcap + k);
      u32 link = bf(y, B_LINK), prev = link & 0xffffu, author = link >> 16;
      u32 round = bf(y, B_ROUND), prev_round = bf(y, B_PREV_ROUND), pp = bf(y, B_PP) & 0xffffu, pp_round = bf(y, B_PP_ROUND);
      u32 epoch = bf(y, B_EPOCH);
      u64 flags = 0;
      // State = DefaultHasher over Vec<(Command, NodeTime)> (simulated_context.rs:51-55): length, then (proposer, index, time)
      Sip13 hs; hs.init(); hs.word(k + 1);
      for (u32 j = 0; j <= k; j++) {
        u32 b = ld(P.off_log + node * P.lcap + j);
        hs.word(blk_author(b)); hs.word(bf(b, B_CMD)); hs.word((u64)(i64)(i32)bf(b, B_TIME));
      }
      u64 state = hs.finish();
      if (prev && prev != y_prev) flags |= 2;
      u64 prev_qc_hash = prev ? qc_prev : record_hash_epoch_id(epoch);
      u64 bh = record_hash_block(author, bf(y, B_CMD), (i64)(i32)bf(y, B_TIME), prev_qc_hash, round, author);
      // vote_committed_state (record_store.rs:237-255): three contiguous rounds commit the grandparent's state
      bool has_cs = prev && pp && round == prev_round + 1 && prev_round == pp_round + 1;
      if (has_cs && pp != y_prev2) flags |= 2;
      u64 cs = has_cs ? state_prev2 : 0;
      // QuorumCertificate_ (record.rs:82-99): epoch_id, round, certified_block_hash, state, committed_state, votes, author
      SipBytes hq; hq.init();
      const char name[] = "QuorumCertificate_::";
      for (u32 i = 0; i < sizeof(name) - 1; i++) hq.byte((u32)name[i]);
      hq.u64le(epoch); hq.u64le(round); hq.u64le(bh); hq.u64le(state); hq.option(has_cs, cs);
      u32 votes = 0;
      for (u32 w = 0; w < MW(); w++) votes += popc64(ld(w == 0 ? bfw(y, B_VOTERS) : bfw(y, B_WORDS + 3 * (MW() - 1) + w - 1)));
      hq.uleb(votes);
      for (u32 w = 0; w < MW(); w++)
        for (u32 m = ld(w == 0 ? bfw(y, B_VOTERS) : bfw(y, B_WORDS + 3 * (MW() - 1) + w - 1)); m; m &= m - 1) {
          u64 a = 32 * w + ctz32(m);  // (Author, Signature{author, hash of the vote}) (simulated_context.rs:22-23,259-261)
          hq.u64le(a); hq.u64le(a); hq.u64le(record_hash_vote(epoch, round, bh, state, has_cs, cs, a));
        }
      hq.u64le(author);
      u64 qh = hq.finish();
      if (!votes) { flags |= 1; qh = 0; }
      out[4 * (size_t)k] = bh; out[4 * (size_t)k + 1] = state; out[4 * (size_t)k + 2] = qh; out[4 * (size_t)k + 3] = votes | (flags << 32);
      y_prev2 = y_prev; y_prev = y; state_prev2 = state_prev; state_prev = state; qc_prev = qh;
    }
    return nc;
  }

  // ---- Simulator::new (simulator.rs:200-250) + NodeState::make_initial_state (node.rs:87-114) ----
  LBFT_HD void init(u64 seed) {
    for (u32 w = 0; w < I_WORDS; w++) st(w, 0);
    clock = 0; stamp = 0; qlen = 0; nblocks = 0; fault = 0; maxq = 0; maxsnap = 0;
    ev0 = ev1 = ev2 = ev3 = 0; n_fold = 0; n_upd = 0; cont = 0;
    sp_idx = 0; sp_s1 = 0; sp_meta = 0; sp_nx = 0; cur_h = 0;
    blk_cache_reset();
    snap_free = P.scap;
    snap_mask = P.scap >= 64 ? ~0ULL : ((1ULL << P.scap) - 1);
    last_node = 0; vd_time = 0xffffffffu; vd_stamp = 0;
    cal_cursor = 0; cal_free = 0; cal_bump = 0;  // (the calendar's head / tail / bitmap rows are zeroed by the host)
    if (P.rcap) {
      for (u32 k = 0; k < NN() * P.rcap; k++) st(P.off_trace + k, 0xffffffffu);
      for (u32 k = 0; k < NN(); k++) st(P.off_trace + NN() * P.rcap + k, 0);
    }
    for (u32 s = 0; s < P.scap; s++) { st(OFFSFREE() + s, P.scap - 1 - s); st(OFFSREF() + s, 0); }  // (large networks: the count lives in the slot, written when it is taken)
    for (u32 k = 0; k < NN() * P.ecap * P.rarch_words; k++) st(P.off_rarch + k, 0);  // (an unused archive entry reads as "no store": current_round 0)
    rng.seed(seed);
    for (u32 node = 0; node < NN(); node++) {
      for (u32 f = 0; f < NWORDS(); f++) nfms(node, f, 0);
      nfms(node, NF_CUR_ROUND, 1);
      nfms(node, NF_PM_LEADER, LBFT_NO_LEADER);
      nfms(node, NF_LAST_TIMER_T, 0xffffffffu);
      i64 startup = 0 + sample_delay() + 1;
      // A node whose startup lies beyond max_clock never runs its own timer, but it still receives notifications and may
      // even propose: its NodeTime (clock - startup, negative) goes into its blocks, so the startup time must be exact
      // (clamping it to max_clock + 1 changed the committed commands' times under long-tailed delays).  Only the 32-bit
      // storage bounds it.
      if (startup > 0x3fffffffLL) { startup = 0x3fffffffLL; fault |= F_INTERNAL; }  // (not representable: say so instead of diverging silently)
      nfms(node, NF_STARTUP, (u32)(i32)startup);
      nfms(node, NF_IGNORE_UNTIL, (u32)(i32)(startup - 1));
      push_event(startup, 3, node, 0, 0);
    }
    store_scalars(false);
  }

  // ---- DataWriter::update_round_number (data_writer.rs:34-50), called by loop_until for every popped event with
  // the event's own scheduled time (simulator.rs:393-396).  Only the node of the previous event can have entered
  // a new round since the last call, so one node is examined instead of all. ----
  LBFT_HD void trace_round_switch(u32 node, i32 event_time) {
    u32 ar = nfm(node, NF_PM_ROUND);
    u32 mw = P.off_trace + NN() * P.rcap + node;
    if (ar > ld(mw)) {
      st(mw, ar);
      // (the reference writes rounds 0..max_round exclusive, data_writer.rs:74-75: a node AT round rcap loses nothing)
      if (ar < P.rcap) st(P.off_trace + node * P.rcap + ar, (u32)event_time);
      else if (ar > P.rcap) fault |= F_TRACE_OVERFLOW;
    }
  }

  // ---- Simulator::loop_until (simulator.rs:380-475); returns true when the queue drained ----
  // One event = step_begin (pop, the node's rows, the event's handler, update_node, process_node_actions, the messages the
  // scalar send loop handles) + step_end (write-back); the cooperative kernels run coop_bulk between the two.
  struct StepCtx { u32 node, sender, kind; i32 t_event; bool do_update; };
  // (fkey, fbest): class 8 only -- the pop's scan was done by the whole wavefront (run_popc / coop_find): smallest LDS-resident key, its slot
  LBFT_HD bool step_begin(StepCtx& c, u64 fkey = 0, u32 fbest = 0) {  // false: the queue is empty
    {
      i32 t; u32 kind, meta;
      // A response that spans several epochs (quirks bit 0) is one event but several steps: between two epochs the reference runs
      // process_commits + update_tracker (handle_response_epoch), which this loop does at update_node's site, writes the node
      // back and comes round again -- the event loop itself is the back edge (a loop inside the step kept 50 more
      // registers alive across it).
      const bool resumed = q1() && cont != 0;
      if (resumed) { t = clock; kind = 2; meta = ld(I_CONT_META); }
      else {
        if (POPC) {  // (the scan was done by the whole wavefront: run_popc)
          if (qlen == 0) return false;
          pop_take(fkey, fbest, t, kind, meta);
        } else if (!pop_event(t, kind, meta)) return false;
        LBFT_MARK(0);
        LBFT_COUNT(30);
        if (tracing()) trace_round_switch(last_node, t);
      }
      i32 t_event = t;
      if (t > clock) clock = t;
      u32 node = meta & 0xffu, sender = (meta >> 8) & 0xffu, slot = meta >> 16;
      if (!resumed) LBFT_HOOK_POP(t_event, kind, node, sender);
      if (!LEAN && !(C0 && LBFT_C0_NO_TRACE_STATE)) last_node = node;
      // One shared site for the node-row burst (and, for a notification, its snapshot words in the same
      // burst), one for update_node + process_node_actions: lanes of a wavefront that handle different
      // event kinds issue their loads together instead of one serialized round trip per kind.
      bool do_update = true, sync = false, more = false;
      SendPlan sp;
      sp.response = 0; sp.resp_slot = 0; sp.sync = 0; sp.sync_stamp = 0; sp.sync_epoch = 0; sp.sync_certs = 0; sp.have_actions = 0;
      // (reference semantics, quirk Q1: a request is answered by the requester itself with a payload-free response -- nothing of the node is read or
      // written, so its rows are not fetched: a third of a large network's events, three lines each; class 0 keeps the burst, its lanes run in lockstep)
      if (C0 || kind != 1 || q1()) begin_node((q1() && kind == 1) ? sender : node);  // Q1 fixed: a request is processed on the peer it was sent to
      Snap sn;
      LBFT_UNROLL
      for (u32 f = 0; f < S_FIXED_WORDS; f++) sn.w[f] = 0;
      sn.refs = 0; sn.to_hcbr[0] = sn.to_hcbr[1] = sn.to_hcbr[2] = sn.to_hcbr[3] = 0;
      if (kind == 0) sn = load_snapshot(slot);
      LBFT_DRAIN_VMEM();
      LBFT_MARK(1);
      if (kind == 3) {  // UpdateTimerEvent (simulator.rs:403-415)
        LBFT_STAT(0);
        ev3 += 1 + slot;  // slot > 0: a materialised group of folded duplicates (see process_node_actions)
        if ((u32)clock == nf(node, NF_LAST_TIMER_T)) {  // folded duplicates of this timer
          if (!LEAN && !(C0 && LBFT_C0_NO_TRACE_STATE) && nf(node, NF_TIMER_DUPS) != 0) {  // their pops follow, interleaved by stamp with the other timers of this time (round trace only)
            u32 ds = nf(node, NF_DUP_STAMP) + 1;
            vd_stamp = (vd_time == (u32)clock && vd_stamp > ds) ? vd_stamp : ds;
            vd_time = (u32)clock;
          }
          ev3 += nf(node, NF_TIMER_DUPS);
          nfs(node, NF_TIMER_DUPS, 0);
          nfs(node, NF_LAST_TIMER_T, 0xffffffffu);
        }
        if (clock <= (i32)nf(node, NF_IGNORE_UNTIL)) {  // cancelled timer
          LBFT_STAT(1);
          do_update = false;
          end_node(node);
        }
        LBFT_MARK(2);
      } else if (kind == 0) {  // DataSyncNotifyEvent (simulator.rs:416-440)
        ev0++; LBFT_STAT(2);
        sync = handle_notification(node, sender, slot, sn);
        snap_release(slot, sn.refs, sn.w[S_EPOCH]);
        if (q1()) { sp.sync_epoch = nf(node, NF_EPOCH); sp.sync_certs = nf(node, NF_HCC_BLK) | (nf(node, NF_HQC_BLK) << 16); }  // the request is created now (data_sync.rs:170-176)
        LBFT_MARK(3);
      } else if (kind == 1) {  // DataSyncRequestEvent (simulator.rs:441-453)
        ev1++; LBFT_STAT(3);
        if (q1()) {
          // handle_request on the peer `sender` (data_sync.rs:183-207): its store now, plus what the request said
          u32 qb = sfw(slot, 0);
          u32 req_epoch = ld(qb + S_EPOCH), req_certs = ld(qb + S_CERTS);
          if (refpack()) { snap_release(slot, req_epoch >> 16, req_epoch & 0xffffu); req_epoch &= 0xffffu; }
          else snap_release(slot);
          i32 rs = snap_alloc();
          if (rs >= 0) {
            u32 rb = sfw((u32)rs, 0);
            // (cooperative kernels: the up to 2n hcbr words of the peer's timeouts are copied by all lanes of the wavefront right
            // after this step's sends, lane = author, instead of eight per round trip here: bit 2 of `bulk`, the slot above it)
            write_store_snapshot(sender, rb, coop());
            if (coop()) bulk_copy = 1u | ((u32)rs << 8);
            st(sqw(rb, 0), req_epoch); st(sqw(rb, 1), req_certs);
          }
          sp.resp_slot = (u32)rs;
        }  // (reference semantics, Q1: answered by the requester itself; the response carries nothing insertable)
        sp.response = 1;
        do_update = false;
        LBFT_MARK(4);
      } else {  // DataSyncResponseEvent (simulator.rs:454-466): under Q1 handle_response inserts nothing
        if (!resumed) { ev2++; LBFT_STAT(4); }
        if (q1()) {
          // (fetched here, one round trip after the node rows: carried through the node-row burst these five words cost the
          // two-wavefront kernel 36 more spilled registers)
          Resp rp = load_response(slot);
          u32 e = resumed ? cont - 1u : rp.req_epoch;
#if defined(LBFT_HOST_STATS) && !defined(__HIPCC__)
          if (!resumed) {  // (tests/tools/run_stats.cpp, analysis only: how many responses would a conservative "nothing to insert, nothing to update" predicate let through?)
            i64 nx_;
            const bool quiet_ = update_is_noop(node, (i64)clock - (i64)(i32)nf(node, NF_STARTUP), nx_);
            const bool inert_ = response_is_inert(node, rp, slot);
            if (inert_) LBFT_STAT(33);
            if (inert_ && quiet_) LBFT_STAT(38); else LBFT_STAT(39);
            static thread_local unsigned streak_ = 0, streak_t_ = ~0u;
            i64 tn_ = (i64)((u64)nx_ + (u64)(i64)(i32)nf(node, NF_STARTUP));
            if (tn_ < (i64)clock + 1) tn_ = (i64)clock + 1;
            const bool folds_ = tn_ > (i64)P.max_clock || (u32)tn_ == nf(node, NF_LAST_TIMER_T);
            const bool ok_ = inert_ && quiet_ && folds_;
            if (streak_t_ != (unsigned)clock || !ok_) { if (streak_ >= 2) { LBFT_STAT(57); LBFT_STATN(56, streak_); } if (streak_ >= 4) LBFT_STATN(63, streak_); streak_ = 0; }
            streak_t_ = (unsigned)clock;
            if (ok_) streak_++;
          }
#endif
          more = handle_response_epoch(node, sender, slot, e, rp);
          if (more) { LBFT_STAT(46); cont = e + 1u; if (!resumed) st(I_CONT_META, meta); }
          else { cont = 0; snap_release(slot); }
        }
        LBFT_MARK(5);
      }
      Actions a;
      a.next = LBFT_NEVER; a.send_to = -1; a.broadcast = false; a.query_all = false;
      if (do_update) {
#if !defined(LBFT_NO_EXEC_COUNTERS)
        if (!more) n_upd++;
#endif
        a = node_update(node, more);  // (more: only the commit / tracker tail, nothing scheduled or sent; the event goes on at the next step)
        if (!more) {
          if (sync) { sp.sync = 1; sp.sync_stamp = stamp++; }  // the request is scheduled before the timer (simulator.rs:424-440)
          LBFT_MARK(11);
          process_node_actions(node, a);
          sp.have_actions = 1;
        }
      }
#if defined(LBFT_HOST_STATS) && !defined(__HIPCC__)
      // (tests/tools/run_stats.cpp, analysis only: how many notifications leave their node as it was -- nothing inserted, no request back, a no-op update?)
      if (kind == 0) {
        static thread_local unsigned streak_ = 0, streak_t_ = ~0u;
        const bool inert_ = !sync && do_update && (cdirty & ~1u) == 0 && axdirty == 0 && a.send_to < 0 && !a.broadcast && !a.query_all;
        LBFT_STAT(43);
        if (streak_t_ != (unsigned)t_event || !inert_) { if (streak_ >= 4) LBFT_STATN(58, streak_); if (streak_ >= 8) LBFT_STATN(59, streak_); streak_ = 0; }
        streak_t_ = (unsigned)t_event;
        if (inert_) { LBFT_STAT(47); streak_++; }
      }
#endif
      send_loop(node, sender, sp, a);
      c.node = node; c.sender = sender; c.kind = kind; c.t_event = t_event; c.do_update = do_update;
    }
    return true;
  }
  LBFT_HD void step_end(const StepCtx& c) {
    if (c.do_update) {
      LBFT_DRAIN_VMEM();
      LBFT_MARK(14);
      end_node(c.node);
      LBFT_MARK(15);
      LBFT_DRAIN_VMEM();
      LBFT_MARK(17);
    }
    // folded duplicate timers of this scheduled time still pop after this timer in the reference
    if (tracing() && c.kind == 3 && (u32)c.t_event == vd_time && ev_stamp < vd_stamp) trace_round_switch(c.node, c.t_event);
  }
  LBFT_HD bool run() {
    u32 steps = 0;
    u32 max_steps = P.max_steps ? P.max_steps : 0xffffffffu;
    LBFT_PIN_VGPR(max_steps);
    bulk = 0;
    for (;;) {
      if (steps >= max_steps) return false;
      StepCtx c;
      if (!step_begin(c)) return true;
      steps++;
      // (a kernel class with cooperative bulk sends that is run lane-per-network -- the generic read-back class, the host
      // model's scalar mode -- never defers a list: coop() is false outside run_coop's classes)
      step_end(c);
      LBFT_STEP_DONE();
    }
  }
  // Kernel class 0: EVERY lane of the wavefront runs the loop and takes part in the scan of the event queues (coop_find); the lanes
  // that carry a network (`leader`) then execute its event.  `kw`: the wavefront's LDS key array.
  LBFT_HD bool run_popc(bool leader, const u64* kw) {
    u32 steps = 0;
    u32 max_steps = P.max_steps ? P.max_steps : 0xffffffffu;  // (>= 1)
    if (!WUNI) LBFT_PIN_VGPR(max_steps);
    bool go = leader && qlen != 0, drained = true;
    bulk = 0;
    // (tested at the bottom: with the exit in the middle of the body the compiler kept two copies of the loop-carried state -- the one the
    // exit uses and the one the step updates -- and moved ~28 registers from one to the other and back in every iteration)
#if defined(__HIP_DEVICE_COMPILE__)
    bool any = __ballot(go) != 0;
#else
    bool any = go;
#endif
    while (any) {
      u64 fkey; u32 fbest;
      if (PAIR) coop_find_cols(kw, go ? qlen : 0u, fkey, fbest);
      else coop_find(kw, fkey, fbest);
      if (go) {
        StepCtx c;
        step_begin(c, fkey, fbest);
        steps++;
        step_end(c);
        LBFT_STEP_DONE();
        if (steps >= max_steps) { go = false; drained = false; }
        else if (qlen == 0) go = false;
      }
#if defined(__HIP_DEVICE_COMPILE__)
      any = __ballot(go) != 0;
#else
      any = go;
#endif
    }
    return drained;
  }
  // The same loop for the kernels whose lanes cooperate (COOP): EVERY lane of the wavefront runs it; `leader` lanes carry a
  // network and execute the events, the other lanes only take part in the bulk sends of the leaders' networks.
  LBFT_HD bool run_coop(bool leader) {
    u32 steps = 0;
    u32 max_steps = P.max_steps ? P.max_steps : 0xffffffffu;
    bool go = leader, drained = true;
    bulk = 0; bulk_node = 0; bulk_copy = 0; ntf_skip = 0; ntf_done = 0;
    coop_on = true;
    for (;;) {
      if (go && steps >= max_steps) { go = false; drained = false; }
#if defined(__HIP_DEVICE_COMPILE__)
      if (__ballot(go) == 0) break;
#else
      if (!go) break;
#endif
      StepCtx c;
      c.node = 0; c.sender = 0; c.kind = 0; c.t_event = 0; c.do_update = false;
      bulk = 0; bulk_copy = 0;
      bool act = go;
      if (act && P.ring_topup) {  // the generator runs ahead of the consumers in every network of the wavefront at once (see RngT)
        u32 room = rng.ring_room();
        rng.ring_fill(room < P.ring_topup ? room : P.ring_topup);
      }
      if (REQRUN && coop()) {  // a run of >= 2 requests at the head of a network's open bucket: the whole wavefront takes it (coop_requests)
        bool is_req = false, is_rsp = false, is_ntf = false;
        if (act && qlen != 0 && !(q1() && cont != 0)) {  // (a response still going through its epochs comes first: step_begin resumes it)
          cal_open();
          const u32 in_chunk = ((sp_nx >> 6) == (cur_h >> 6) ? (sp_nx & 63u) : LBFT_CAL_CE) - (cur_h & 63u) + 1u;
          is_req = (sp_idx & 3u) == 2u && in_chunk >= LBFT_RUN_MIN && max_steps - steps >= LBFT_RUN_MIN;
          is_rsp = rsp_runs() && (sp_idx & 3u) == 1u && in_chunk >= LBFT_RUN_MIN && max_steps - steps >= LBFT_RUN_MIN;
          is_ntf = NTFRUN && P.lpw <= 32u && (sp_idx & 3u) == 3u && in_chunk >= LBFT_NTF_MIN && max_steps - steps >= LBFT_NTF_MIN && ntf_skip == 0;
        }
#if defined(__HIP_DEVICE_COMPILE__)
        unsigned long long rq = __ballot(is_req);
#else
        unsigned long long rq = is_req ? 1ULL : 0ULL;
#endif
        while (rq) {
          u32 k = ctz64(rq);
          rq &= rq - 1;
          coop_requests(k, LBFT_UNI(max_steps - steps, k));
        }
        if (is_req) { steps += req_done; act = false; LBFT_RUNCNT(0, req_done); }
        if (rsp_runs()) {  // ... and a run of responses whose update_node is a no-op (coop_responses)
#if defined(__HIP_DEVICE_COMPILE__)
          unsigned long long rs = __ballot(is_rsp);
#else
          unsigned long long rs = is_rsp ? 1ULL : 0ULL;
#endif
          while (rs) {
            u32 k = ctz64(rs);
            rs &= rs - 1;
            coop_responses(k, LBFT_UNI(max_steps - steps, k));
          }
          if (is_rsp && rsp_done) { steps += rsp_done; act = false; LBFT_RUNCNT(1, rsp_done); }  // (0: the head event's update does something -- the ordinary step takes it)
        }
        if (NTFRUN) {  // ... and the runs of notifications that leave their nodes as they were, of all networks at once (coop_notifications)
#if defined(__HIP_DEVICE_COMPILE__)
          const bool any_ntf = __ballot(is_ntf) != 0;
#else
          const bool any_ntf = is_ntf;
#endif
          if (any_ntf) coop_notifications(is_ntf, max_steps - steps);
          if (is_ntf && ntf_done) { steps += ntf_done; act = false; LBFT_RUNCNT(2, ntf_done); }
          else if (act && ntf_skip) ntf_skip--;  // (an event seen not to be inert takes its ordinary step)
        }
      }
      if (act) {
        if (!step_begin(c)) { go = false; act = false; } else steps++;
      }
#if defined(__HIP_DEVICE_COMPILE__)
      unsigned long long need = __ballot(act && (bulk | bulk_copy) != 0);
#else
      unsigned long long need = (act && (bulk | bulk_copy) != 0) ? 1ULL : 0ULL;
#endif
      while (need) {
        u32 k = ctz64(need);
        need &= need - 1;
        u32 bk = LBFT_UNI(bulk, k);
        if (bk & 1u) coop_bulk(k, 0);
        if (bk & 2u) coop_bulk(k, 1);
        if (q1()) {  // a request answered in lane k: the peer (the node in its cache) is the event's sender
          u32 bc = LBFT_UNI(bulk_copy, k);
          if (bc & 1u) coop_copy_hcbr_to(k, LBFT_UNI(lane4, k), LBFT_UNI(c.sender, k), sfw(bc >> 8, 0));
        }
      }
      if (act) step_end(c);
    }
    return drained;
  }
};

typedef SimT<K_GENERIC> Sim;
// The class lbft_k_run (and the host model) executes a batch with.
inline int sim_class(const Params& p) {
  if (p.n > 32) return K_LARGE;
  bool small = p.n <= 16 && !p.qheap && !p.equiv && !p.rcap && !p.drop_ppm && !p.part_size && !(p.quirks & 1u);
  bool fits_packed_queue = p.max_clock < (1 << LBFT_QP_TIME_BITS) && p.scap <= 256;  // one-word queue entries
  return small && fits_packed_queue ? K_SMALL : K_MID;
}

// Does a class-0 batch qualify for the kernel with the headline network fixed at compile time (SimT<K_HEADLINE>)?
inline bool sim_quad(const Params& p) {
  return sim_class(p) == K_SMALL && p.n == 4 && p.unit_weights && p.delay_model == 0 && p.scap <= 64 && p.rot == 0 && p.rarch_words == 0 &&
         LBFT_C0_IMAJOR && p.off_node == I_WORDS && p.node_words == NF_FIXED_WORDS + 8u && p.snap_words == S_FIXED_WORDS + 8u &&
         p.blk_words == B_WORDS;
}
// Does a class-2 / class-1 batch qualify for the lean kernel of its class (SimT<K_LARGE_LEAN> / SimT<K_MID_LEAN>)?
inline bool sim_lean_features(const Params& p) { return (!(p.quirks & 1u) || (LBFT_LEAN_Q1 && p.n > 32)) && !p.rcap && !p.drop_ppm && !p.part_size; }
inline bool sim_lean(const Params& p) { return sim_class(p) == K_LARGE && sim_lean_features(p); }
inline bool sim_lean_q1(const Params& p) { return sim_lean(p) && (p.quirks & 1u) != 0; }  // ... SimT<K_LARGE_EXCHANGE> instead of SimT<K_LARGE_LEAN>
inline bool sim_lean1(const Params& p) { return sim_class(p) == K_MID && sim_lean_features(p); }

// Row layout for a batch; fills the offset fields of `p` and returns words per instance.  Accumulated in 64 bits: a tile is
// addressed with 32-bit byte offsets (boff(): row << rsh, at most 8), so a layout is only usable while it stays below 2^24 rows; the
// caller rejects larger ones (layout_fits) instead of letting row offsets wrap.
#define LBFT_CAL_CE_LAYOUT 31u  // (= LBFT_CAL_CE inside SimT)
inline u32 node_words_used(const Params& p) { return NF_FIXED_WORDS + 2 * p.n + 4 * ((p.n + 31) / 32 - 1); }
inline u32 snap_words_used(const Params& p) { return S_FIXED_WORDS + 2 * p.n + 2 * ((p.n + 31) / 32 - 1) + ((p.quirks & 1u) ? 2 : 0); }
inline u32 layout_tile_width(const Params& p) { return sim_class(p) == K_SMALL ? (LBFT_C0_IMAJOR ? 1u : 64u) : sim_class(p) == K_MID ? 64u : 1u; }
inline u64 compute_layout(Params& p) {
  u64 w = I_WORDS;
  p.mw = (p.n + 31) / 32;
  // large networks (instance-major rows): every node row starts on a 128-byte line, so that the 41 + 4 (mw - 1) words an event stages are two lines,
  // not two or three (round 6; the instance's first word is line-aligned as well: total_words below)
  const bool align_rows = p.mw > 1;
  if (align_rows) w = (w + 31) & ~(u64)31;
  p.off_node = (u32)w; p.node_words = NF_FIXED_WORDS + 2 * p.n + 4 * (p.mw - 1);
  if (align_rows) p.node_words = (p.node_words + 31u) & ~31u;
  w += (u64)p.n * p.node_words;
  p.snap_words = S_FIXED_WORDS + 2 * p.n + 2 * (p.mw - 1) + ((p.quirks & 1u) ? 2 : 0);  // + the request's (epoch, certificates)
  p.blk_words = B_WORDS + 4 * (p.mw - 1);  // + extension words (nodes / authors >= 32) of KNOWN, QC, PEND and VOTERS
  // class 0, instance-major: notification snapshots and the block pool follow the nodes, the queue's spill rows come last -- the first
  // word of every hot region then depends on num_nodes and snapshot_capacity alone, which the kernel with the headline network fixed at
  // compile time (SimT<K_HEADLINE>) turns into immediates
  const bool hot_first = LBFT_C0_IMAJOR && LBFT_C0_HOT_FIRST && sim_class(p) == K_SMALL;
  auto snaps_blocks = [&]() {
    p.off_snap = (u32)w; w += (u64)p.scap * p.snap_words;
    p.off_snap_ref = (u32)w; w += p.scap;
    p.off_snap_free = (u32)w; w += p.scap;
    p.off_blk = (u32)w; w += (u64)p.bcap * p.blk_words;
  };
  if (hot_first) snaps_blocks();
  // Calendar queue (round 6): a pool of 32-word chunks (31 event metas + a link word each) in the rows off_qmeta, the stack of freed chunks in the rows
  // off_qlo, no key / link rows.  An entry costs 4.13 bytes; the pool holds qcap entries plus one partly filled tail chunk per bucket that can be
  // non-empty at a time (events only ever sit in the ~delay-spread many time slots ahead of the clock; a margin of qcap / 32 chunks, at least 256) -- running out raises
  // F_QUEUE_OVERFLOW like a full queue.
  p.cal_chunks = 0;
  if (p.qcal) {
    u32 spare = p.qcap / 32u;
    p.cal_chunks = (p.qcap + LBFT_CAL_CE_LAYOUT - 1u) / LBFT_CAL_CE_LAYOUT + (spare < 256u ? 256u : spare);
    if (align_rows) w = (w + 31) & ~(u64)31;  // chunks are lines
  }
  p.off_qhi = (u32)w; w += p.qcal ? 0 : p.qcap;
  p.off_qlo = (u32)w; w += p.qcal ? p.cal_chunks : p.qcap;
  if (p.qcal && align_rows) w = (w + 31) & ~(u64)31;
  p.off_qmeta = (u32)w; w += p.qcal ? (u64)p.cal_chunks * 32u : p.qcap;
  p.cal_buckets = p.qcal ? ((u32)p.max_clock + 1u) * 4u : 0;
  p.off_cal_head = (u32)w; w += 2ULL * p.cal_buckets;  // (head, tail) word pairs: a bucket that becomes non-empty writes both -- one line
  p.off_cal_bm = (u32)w; w += (p.cal_buckets + 31) / 32 + (p.qcal ? 1 : 0);
  if (!hot_first) snaps_blocks();  // (zero_calendar relies on head / tail / bitmap being the rows right before the snapshots here)
  p.off_log = (u32)w; w += (u64)p.n * p.lcap;
  p.off_list = (u32)w; w += p.n > 16 ? p.n : 0;
  p.off_trace = (u32)w; w += p.rcap ? (u64)p.n * p.rcap + p.n : 0;
  p.off_arch = (u32)w; w += (p.quirks & 1u) ? (u64)p.n * p.ecap * p.snap_words : 0;
  if (p.rarch_words) p.rarch_words = node_words_used(p);
  p.off_rarch = (u32)w; w += (u64)p.n * p.ecap * p.rarch_words;
  p.off_sync = (u32)w; w += (p.quirks & 1u) ? p.bcap : 0;
  p.off_ring = (u32)w; w += 2ULL * p.ring;
  if (align_rows) w = (w + 31) & ~(u64)31;
  p.total_words = w > 0xffffffffULL ? 0xffffffffu : (u32)w;
  p.qpack = sim_class(p) == K_SMALL ? 1u : 0u;
  return w;
}
inline bool layout_fits(u64 total_words) { return total_words < (1ULL << 24); }

}  // namespace lbft

#endif  // LBFT_CORE_H
