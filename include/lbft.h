/* lbft.h -- C ABI of the MI355X-native batched LibraBFTv2 discrete-event simulator (liblbft_hip.so).
 *
 * Drop-in boundary for the simulation hot path of novifinancial/librabft_simulator.  Each entry point
 * names the reference interface it replaces (paths relative to the reference repository).  All
 * functions are synchronous, return 0 (LBFT_OK) or a negative LBFT_ERR_* code, never abort the
 * process and never call back into the caller.  A batch handle owns all of its host and device memory
 * and is not thread-safe (one host thread per GPU/batch, as the reference is single-threaded).
 *
 * There is no CPU fallback: every function that needs the GPU fails with LBFT_ERR_HIP when no
 * gfx950 device is usable.
 */
#ifndef LBFT_H
#define LBFT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LBFT_OK 0
#define LBFT_ERR_INVALID (-1)     /* bad argument (NULL pointer, num_nodes out of range, max_clock >= 2^31-1, ...) */
#define LBFT_ERR_HIP (-2)         /* HIP runtime error / no device; see lbft_last_error() */
#define LBFT_ERR_UNSUPPORTED (-3) /* unknown quirks bits or num_nodes > LBFT_MAX_NODES_SUPPORTED on this kernel family */
#define LBFT_ERR_STATE (-4)       /* call order violated (e.g. results requested before lbft_batch_run_until) */
#define LBFT_ERR_FAULT (-5)       /* the run finished but >= 1 instance raised a sticky fault (capacity overflow or an
                                     invariant on which the reference itself would have panicked); see lbft_batch_faults */

#define LBFT_MAX_NODES_SUPPORTED 128

/* Per-instance sticky fault bits (lbft_batch_faults). */
#define LBFT_FAULT_QUEUE_OVERFLOW (1u << 0)
#define LBFT_FAULT_SNAPSHOT_OVERFLOW (1u << 1)
#define LBFT_FAULT_BLOCK_OVERFLOW (1u << 2)
#define LBFT_FAULT_LOG_OVERFLOW (1u << 3)
#define LBFT_FAULT_BALLOT_OVERFLOW (1u << 4)
#define LBFT_FAULT_DURATION_TABLE (1u << 5)
#define LBFT_FAULT_COMMIT_UNKNOWN_STATE (1u << 6) /* simulated_context.rs:163-166 would panic */
#define LBFT_FAULT_COMMIT_NOT_SUCCESSOR (1u << 7) /* simulated_context.rs:172-174 would panic */
#define LBFT_FAULT_STAMP_OVERFLOW (1u << 8)
#define LBFT_FAULT_INTERNAL (1u << 9)
#define LBFT_FAULT_TRACE_OVERFLOW (1u << 11)
#define LBFT_FAULT_EPOCH_OVERFLOW (1u << 12) /* quirks bit 0: record-store archive / response scratch exhausted */

/* Simulation parameters: the arguments of Simulator::new (bft-lib/src/simulator.rs:200-208),
 * RandomDelay::new (:99-106), SimulatedContext::new (bft-lib/src/simulated_context.rs:86-96) and
 * NodeConfig (librabft-v2/src/node.rs:76-81), i.e. the CLI flags of librabft-v2/src/main.rs:73-140. */
typedef struct lbft_config {
  uint32_t num_nodes;             /* --nodes */
  uint32_t delay_model;           /* 0: LogNormal(mean, variance) as the reference; 1: uniform integer in [uniform_lo, uniform_hi] (extension) */
  double mean;                    /* --mean */
  double variance;                /* --variance */
  int64_t uniform_lo, uniform_hi; /* delay_model 1 only */
  uint64_t commands_per_epoch;    /* --commands_per_epoch */
  int64_t target_commit_interval; /* --target_commit_interval */
  int64_t delta;                  /* --delta */
  double gamma;                   /* --gamma */
  double lambda;                  /* --lambda */
  uint32_t quirks;                /* 0: reference semantics incl. quirks Q1/Q2.  Bit 1 (value 2): EpochId::previous()
                                     returns id - 1 as intended (fixes Q2, base_types.rs:31-37): notifications forward the previous epoch's
                                     commit certificate (data_sync.rs:84-92), which keeps a network live across epoch changes.  Bit 0
                                     (value 1): a DataSyncRequest is answered by the peer it was sent to, as bft-driver does
                                     (bft-driver/src/core.rs:174-178), instead of by the requester itself (fixes Q1, simulator.rs:446):
                                     responses carry the records the requester lacks (record_store.rs:766-831) and lagging nodes catch
                                     up.  3 = both.  The oracle implements the same modes. */
  uint32_t equivocate_every;      /* extension, 0 = all honest.  k > 0: every node with index % k == 0 is an equivocating
                                     leader: (E1) whenever its pacemaker makes it propose (node.rs:191-201) it proposes TWO
                                     blocks A, B on the same previous QC (two fetches, same NodeTime; B ends up as its current
                                     proposed block, record_store.rs:466-476); (E2) receivers with an even author index get
                                     its notifications with A instead of B as proposed_block; (E3) otherwise it follows the
                                     protocol.  The reference has no Byzantine behaviour; oracle/lbft_oracle.cpp is the spec. */
  const uint64_t* voting_rights;  /* NULL: every node has weight 1 (simulated_context.rs:209-216); else num_nodes weights */
  /* Capacities of the per-instance device structures; 0 = choose from num_nodes / max_clock. */
  uint32_t queue_capacity;    /* pending events with time <= max_clock */
  uint32_t snapshot_capacity; /* notifications in flight */
  uint32_t block_capacity;    /* blocks proposed per instance (<= 65534) */
  uint32_t log_capacity;      /* commits per node */
  /* Lossy network (extension; the reference lists network changes / disconnects as TODO, simulator.rs:25; the oracle is
   * the specification).  Every message Simulator::schedule_network_event would schedule draws its delay as usual, then
   * (L1) if drop_per_million > 0 one more next_u64() draw d decides: lost iff mulhi64(d, 1000000) < drop_per_million;
   * (L2) if partition_size > 0 and partition_start <= clock < partition_end, messages between a node < partition_size
   * and a node >= partition_size are lost.  A lost message consumes its creation stamp and is never queued. */
  uint32_t drop_per_million;
  uint32_t partition_size;
  int64_t partition_start, partition_end;
  /* Epoch reconfiguration (extension; "We do not simulate changes in the voting rights yet", simulated_context.rs:209-216;
   * the oracle is the specification).  EpochReader::configuration(state) gives author i the voting right
   * voting_rights[(i + e * rights_rotation) % num_nodes] with e = read_epoch_id(state); 0 = the same rights in every epoch.
   * The record store of epoch e counts votes / timeouts with these weights and elects its leaders over them
   * (node.rs:331-348, configuration.rs:65-75).  Meaningful with commands_per_epoch small enough to change epochs and
   * quirks = 3 (with the reference's quirks a network stalls at its first epoch change). */
  uint32_t rights_rotation;
} lbft_config;

/* One entry of SimulatedContext::committed_history() (simulated_context.rs:31-35,98-100). */
typedef struct lbft_commit {
  uint64_t proposer; /* Command.proposer = Author(usize) */
  uint64_t index;    /* Command.index */
  int64_t time;      /* NodeTime chosen by the proposer */
} lbft_commit;

/* Aggregate counters of a finished run (sums over the instances of the batch). */
typedef struct lbft_counters {
  uint64_t events[4];         /* processed events by Event::kind (simulator.rs:129-139): notify, request, response, timer */
  uint64_t rng_draws;         /* next_u64 calls on the simulators' RNGs */
  uint64_t rounds;            /* sum over instances of min over nodes of pacemaker.active_round */
  uint64_t commits;           /* sum over instances of min over nodes of committed_history().len() */
  uint64_t events_scheduled;  /* creation stamps handed out (simulator.rs:252-264) */
  uint64_t faulted_instances; /* instances with a non-zero fault word */
  uint64_t max_queue;         /* max over instances of the event-queue high-water mark */
  uint64_t max_snapshots;     /* max over instances of live notification snapshots */
  uint64_t max_blocks;        /* max over instances of proposed blocks */
  uint64_t launches;          /* run-kernel launches of the last run */
  /* What the device executes, as opposed to the reference-equivalent counts above: duplicate UpdateTimerEvents of one (node, time) are
   * folded at scheduling time and only counted in events[3] (simulator.rs:406-410 makes them no-ops), so the number of queue pops is
   * events[0..3] summed minus timers_folded (round trace off); node_updates = ConsensusNode::update_node calls, i.e. the events that
   * load AND write back a node's rows (cancelled timers and requests only load them). */
  uint64_t timers_folded;
  uint64_t node_updates;
} lbft_counters;

typedef struct lbft_batch lbft_batch;

/* Builds n_instances independent simulators, instance i seeded with seeds[i]
 * (replaces n_instances calls of Simulator::new, bft-lib/src/simulator.rs:200-250, with the context
 * factory of librabft-v2/src/main.rs:23-34).  `device` is the HIP device ordinal.  Seeds are copied to
 * the device here; nothing else crosses PCIe until results are read back. */
int lbft_batch_create(const lbft_config* cfg, const uint64_t* seeds, size_t n_instances, int device, lbft_batch** out);

/* Simulator::loop_until(GlobalTime(max_clock), None) for every instance (simulator.rs:380-475), including
 * the initial scheduling done by Simulator::new.  0 <= max_clock < 2^31 - 1.  May be called again after
 * lbft_batch_reset.  Returns LBFT_ERR_FAULT if any instance faulted (results of the others are valid). */
int lbft_batch_run_until(lbft_batch* b, int64_t max_clock);

/* Forgets the results so that lbft_batch_run_until can run the same seeds again (benchmarking). */
int lbft_batch_reset(lbft_batch* b);

/* contexts.iter().map(|c| c.committed_history().len()) (librabft-v2/src/main.rs:47-53): out[inst * num_nodes + node]. */
int lbft_batch_commit_counts(const lbft_batch* b, uint32_t* out);
/* ActiveRound::active_round per node (simulator.rs:86-88, node.rs:170-175): out[inst * num_nodes + node]. */
int lbft_batch_active_rounds(const lbft_batch* b, uint64_t* out);
/* SimulatedContext::committed_history() of one node (simulated_context.rs:98-100); copies min(cap, *len) entries. */
int lbft_batch_committed_history(const lbft_batch* b, size_t inst, uint32_t node, lbft_commit* out, size_t cap, size_t* len);
/* The records behind one node's committed history with the hashes the reference gives them: SmrContext::hash = SipHash-1-3
 * (Rust DefaultHasher) of "Name::" + BCS bytes (smr_context.rs:84-95, simulated_context.rs:238-242).  Entry k describes the
 * Block_ that carried the k-th committed command (record.rs:51-63), the State after executing it (simulated_context.rs:51-55)
 * and the QuorumCertificate_ certifying the block (record.rs:82-99; its Vote_s, record.rs:65-80, in author order).  The event
 * loop never needs these hashes (records are identified structurally); they are recomputed on the device
 * from the block pool, so that simulated state can be related to what a reference node would store or put on the wire
 * (first half).  flags: bit 0 = no QC recorded for the block, bit 1 = internal inconsistency.
 * Copies min(cap, *len) entries. */
typedef struct lbft_record_hash {
  uint64_t block_hash; /* context.hash(&Block_ { command, time, previous_quorum_certificate_hash, round, author }) */
  uint64_t state;      /* State(..) of the ledger after the block's command */
  uint64_t qc_hash;    /* context.hash(&QuorumCertificate_ { epoch_id, round, certified_block_hash, state, committed_state, votes, author }) */
  uint32_t num_votes;  /* votes in the QC */
  uint32_t flags;
} lbft_record_hash;
int lbft_batch_committed_record_hashes(const lbft_batch* b, size_t inst, uint32_t node, lbft_record_hash* out, size_t cap, size_t* len);
/* All histories: out[(inst * num_nodes + node) * cap_per_node + k], first min(len, cap_per_node) entries each. */
int lbft_batch_committed_histories(const lbft_batch* b, lbft_commit* out, size_t cap_per_node);
/* StateFinalizer::last_committed_state() (simulated_context.rs:194-196; State = SipHash-1-3 of the
 * history, :51-55), computed on the device: out[inst * num_nodes + node]. */
int lbft_batch_last_committed_states(const lbft_batch* b, uint64_t* out);
/* ConsensusNode::save_node (librabft-v2/src/node.rs:233-238): bincode::serialize(&NodeState) of one node -- record store with every Block /
 * QuorumCertificate / Vote / Timeout it holds (rebuilt with the reference's BCS + SipHash-1-3 hashes and signatures), pacemaker, tracker --
 * in bincode 1.3's default encoding with HashMaps in ascending key order (the reference's own order is per-process; its load_node,
 * node.rs:211-231, accepts any order).  *len = the image's length; buf == NULL: size query; a buffer smaller than the image is
 * LBFT_ERR_INVALID (*len still holds the size needed).  A node that has changed epoch carries its retired record stores
 * (past_record_stores, node.rs:43,338-340): the device keeps them in full only when lbft_batch_keep_retired_stores(b, 1) was called
 * before the run; otherwise such a node is LBFT_ERR_UNSUPPORTED (nodes still in epoch 0 always work). */
int lbft_batch_save_node(const lbft_batch* b, size_t inst, uint32_t node, void* buf, size_t cap, size_t* len);
/* ConsensusNode::load_node (librabft-v2/src/node.rs:211-231): bincode::deserialize::<NodeState>(image) into the device-resident node
 * (inst, node) -- the inverse of lbft_batch_save_node, any HashMap order accepted.  `node_time` is the reference's guard
 * (node.rs:219-228): an image whose latest_query_all_time / tracker.latest_commit_time / pacemaker.active_round_start_time lies after it
 * is refused with LBFT_ERR_STATE ("refusing to restore saved state from the future") and nothing is written.  Records are identified
 * by their hashes among the records of the instance's block pool: an image saved from this instance (any node, now or earlier), from
 * another batch run with the same seed and configuration, or by the reference / the oracle for the same run loads; an image that names
 * a record the pool does not hold is LBFT_ERR_UNSUPPORTED (and so is an image with retired record stores loaded into a batch created without
 * lbft_batch_keep_retired_stores: they would be dropped silently and the node could not be saved again), one saved under another configuration (nodes, voting rights, NodeConfig) or
 * malformed is LBFT_ERR_INVALID -- in every failing case the node is left untouched.  Written: the NodeState (record store with the
 * node's blocks / certificates / timeouts / votes / election, pacemaker, epoch, voting constraints, commit tracker, retired record
 * stores).  Not written: what the reference keeps outside NodeState -- the simulator's timer bookkeeping and startup time
 * (SimulatedNode, bft-lib/src/simulator.rs:53-59) and the SmrContext (ledger and pending states, simulated_context.rs:75-83), which its
 * load_node receives separately.  Call after a run, in a node-level session, or between bounded launches (lbft_batch_run_steps). */
int lbft_batch_load_node(lbft_batch* b, size_t inst, uint32_t node, const void* image, size_t len, int64_t node_time);
/* Multi-GPU: instances shard over the GPUs with no data-path collective; the run's ONE collective aggregates the
 * throughput counters of all ranks: ONE RCCL ncclAllGather over xGMI of the fourteen counter words of lbft_batch_counters per rank,
 * reduced locally (the eleven additive counters summed, the three high-water marks by maximum: two reduction operators, which one
 * ncclAllReduce cannot mix -- the same single collective the Python host issues through torch.distributed, distributed.py
 * gather_rows).  `nccl_comm` is the caller's ncclComm_t for this batch's device (any
 * host language: the library loads librccl itself, on first use); the collective runs on the batch's stream.  `out` receives the
 * aggregate; launches is this rank's.  Every rank of the communicator must call it. */
int lbft_batch_counters_allgather_reduce(lbft_batch* b, void* nccl_comm, lbft_counters* out);
/* The same call under the name it had through round 4 (it never issued an all-reduce); kept for existing bindings. */
int lbft_batch_counters_allreduce(lbft_batch* b, void* nccl_comm, lbft_counters* out);
/* past_record_stores kept in full: at every epoch change the node's rows (record-store fields, timeouts, votes, election) are copied
 * into an archive entry of the epoch being left -- num_nodes x epochs x one node's rows of device memory per instance.  Call before
 * the batch runs.  Off by default: a run's results never depend on it. */
int lbft_batch_keep_retired_stores(lbft_batch* b, int enable);
/* ... and of one node: contexts[node].last_committed_state() as librabft-v2/tests/simulated_run.rs:57-65 reads it. */
int lbft_batch_last_committed_state(const lbft_batch* b, size_t inst, uint32_t node, uint64_t* out);
/* SimulatedNode::startup_time (simulator.rs:55,217): out[inst * num_nodes + node]. */
int lbft_batch_startup_times(const lbft_batch* b, int64_t* out);
/* Current epoch of each node (librabft-v2/src/node.rs:34): out[inst * num_nodes + node]. */
int lbft_batch_epochs(const lbft_batch* b, uint64_t* out);
int lbft_batch_counters(const lbft_batch* b, lbft_counters* out);
/* out[inst] = sticky fault bits. */
int lbft_batch_faults(const lbft_batch* b, uint32_t* out);
void lbft_batch_destroy(lbft_batch* b);

/* Measurement hooks (bench.py): the HIP stream all kernels of this batch are launched on, the kernel
 * time of the last lbft_batch_run_until measured with hipEvents on that stream, and the HBM footprint. */
void* lbft_batch_stream(const lbft_batch* b);
int lbft_batch_last_run_ms(const lbft_batch* b, float* init_ms, float* run_ms);
size_t lbft_batch_device_bytes(const lbft_batch* b);
/* Sizes behind the roofline arithmetic (bench.py): out[8] = bytes ONE EVENT MOVES of a node's rows (the fixed words + set extension
 * words of a node burst; plus the node's 2n hcbr words where they ride in the burst: lbft_k_run0q), of one queued event, of one
 * notification snapshot, of one block record, HBM bytes per instance, LDS-resident queue slots, lanes per wavefront,
 * kernel size class | heap-queue flag << 8 | calendar-queue flag << 9 | two-wavefronts-per-SIMD ("lean") kernel flag << 10 |
 * cooperative-bulk-send flag << 11 (large networks on the calendar queue: all lanes of a wavefront execute a network's broadcasts) |
 * (lean kernel with the record exchange of quirks bit 0, lbft_k_run2q) << 12 | (small-batch class-0 kernel lbft_k_run0s: the pop's scan
 * of the LDS event queues by all lanes of the wavefront) << 13 | (class-0 kernel with the headline network -- 4 nodes, unit voting rights,
 * log-normal delays -- fixed at compile time, lbft_k_run0q) << 14 | (small-batch kernel with ONE network per wavefront executed as
 * wavefront-uniform code on the scalar unit, lbft_k_run0u -- set together with bit 13) << 15. */
int lbft_batch_layout(const lbft_batch* b, uint32_t* out);
/* Events processed per run-kernel launch (0 = whole simulation in one launch). */
int lbft_batch_set_max_steps(lbft_batch* b, uint32_t max_steps);
/* Tuning: event-queue slots per instance kept in LDS by the run kernel (-1 = as many as the CU's 160 KiB
 * afford, 0 = queue in HBM only); slots beyond it spill to HBM.  Results do not depend on it. */
int lbft_batch_set_lds_queue_slots(lbft_batch* b, int32_t slots);
/* Tuning / tests: 0 forces the binary-heap event queue where the calendar queue would be used (networks outside kernel
 * class 0 with max_clock <= 2047).  Results do not depend on it. */
int lbft_batch_set_calendar_queue(lbft_batch* b, int enabled);
/* Diagnostic builds (-DLBFT_PHASE_TIMERS) only: out[32] = shader cycles per phase of the event loop summed
 * over wavefronts ([30] = wavefront loop iterations, [31] = total cycles) (see Sim::run); LBFT_ERR_UNSUPPORTED in product builds. */
int lbft_batch_phase_cycles(const lbft_batch* b, uint64_t* out);
/* Tuning: how many of a wavefront's 64 lanes carry an instance (0 = auto from the batch size). Results do not depend on it. */
int lbft_batch_set_lanes_per_wavefront(lbft_batch* b, uint32_t lanes);

/* Stepwise execution and checkpoint / resume.  The reference persists every node after every event (save_node:
 * bincode of the whole NodeState, node.rs:211-238, simulator.rs:307-309) and reads it back once (load_node,
 * simulator.rs:221); inside a run that is a semantic no-op, so this library checkpoints at batch granularity instead:
 * the complete device state (nodes, queues, RNGs, blocks, logs) of all instances.
 *   lbft_batch_run_steps: first call = Simulator::new for every instance; every call processes at most `steps`
 *     events per instance (0 = until drained) and reports how many instances still have pending events; when that
 *     reaches 0 the batch is finished and every read-back call works as after lbft_batch_run_until (same results).
 *   lbft_batch_checkpoint_save / _load: between two lbft_batch_run_steps calls; load into a batch created with the
 *     same configuration and seeds count (typically in another process), then keep calling lbft_batch_run_steps. */
int lbft_batch_run_steps(lbft_batch* b, int64_t max_clock, uint32_t steps, uint64_t* unfinished);
size_t lbft_batch_checkpoint_bytes(const lbft_batch* b);
int lbft_batch_checkpoint_save(const lbft_batch* b, void* buf, size_t cap);
int lbft_batch_checkpoint_load(lbft_batch* b, const void* buf, size_t len);

/* DataWriter of the reference (`--create_csv`; bft-lib/src/data_writer.rs, called per popped event from
 * simulator.rs:393-396): records, per node, the GlobalTime of the first popped event at which the node was seen in
 * each round, and counts the non-timer events.  Enable before lbft_batch_run_until (max_rounds = rows kept per
 * node; going past it raises LBFT_FAULT_TRACE_OVERFLOW for that instance).  lbft_batch_round_switches returns the
 * cells of round_switches.txt for one instance: out[round * num_nodes + node], INT64_MIN = empty cell, rows for
 * round < min(*max_round, cap_rounds); *messages = the value of number_of_messages.txt. */
int lbft_batch_enable_round_trace(lbft_batch* b, uint32_t max_rounds);
int lbft_batch_round_switches(const lbft_batch* b, size_t inst, int64_t* out, size_t cap_rounds, uint64_t* max_round, uint64_t* messages);

/* ---- Node-level interface: the reference's trait surface for ONE node of ONE instance, without the event loop
 * (bft-lib/src/interfaces.rs:12-86), so that record-store / pacemaker scenarios can be replayed step by step
 * (librabft-v2/src/unit_tests/record_store_tests.rs) and a host-side driver other than the batch simulator (the
 * role bft-driver/src/core.rs:125-198 plays for the real network) can own time and message delivery.  Each call
 * launches a one-lane kernel on the batch's stream and blocks.  Usage: lbft_batch_create, lbft_batch_manual_begin,
 * then any sequence of lbft_node_* calls; lbft_batch_manual_finalize makes the batch-level read-back calls
 * (commit counts, histories, States) available. ---- */
typedef struct lbft_actions { /* NodeUpdateActions (interfaces.rs:12-21) */
  int64_t next_scheduled_update; /* NodeTime; INT64_MAX = never */
  uint64_t should_send[2];       /* bit set of authors */
  uint32_t should_broadcast, should_query_all;
} lbft_actions;
typedef struct lbft_node_view { /* RecordStoreState / PacemakerState / NodeState scalars (record_store.rs:93-119, pacemaker.rs:60-77, node.rs:28-45) */
  uint64_t epoch_id, current_round, highest_quorum_certificate_round, highest_timeout_certificate_round,
      highest_committed_round, active_round, latest_voted_round, locked_round, commit_count;
  uint32_t active_leader; /* UINT32_MAX = None */
  uint32_t election;      /* 0 ongoing, 1 won, 2 closed (record_store.rs:125-134) */
  uint32_t num_current_timeouts, num_current_votes, has_proposed_block, has_timeout_certificate;
} lbft_node_view;
/* NodeState::make_initial_state for every node of every instance (node.rs:87-114), no events are processed. */
int lbft_batch_manual_begin(lbft_batch* b, int64_t max_clock);
int lbft_batch_manual_finalize(lbft_batch* b);
/* ConsensusNode::update_node(&mut self, &mut Context, clock: NodeTime) -> NodeUpdateActions (interfaces.rs:37-49, node.rs:240-304) */
int lbft_node_update(lbft_batch* b, size_t inst, uint32_t node, int64_t node_time, lbft_actions* out);
/* DataSyncNode::create_notification (interfaces.rs:62-64, data_sync.rs:82-111): *handle names the notification
 * until lbft_node_release_notification; it may be delivered to any number of receivers. */
int lbft_node_create_notification(lbft_batch* b, size_t inst, uint32_t node, uint32_t* handle);
/* DataSyncNode::handle_notification (interfaces.rs:72-76, data_sync.rs:113-177); *should_sync != 0 when the
 * reference would return Some(Request). */
int lbft_node_handle_notification(lbft_batch* b, size_t inst, uint32_t receiver, uint32_t sender, uint32_t handle, uint32_t* should_sync);
int lbft_node_release_notification(lbft_batch* b, size_t inst, uint32_t handle);
/* The other half of the DataSyncNode trait (interfaces.rs:54-86).  In batches created with quirks bit 0 (the record-exchange
 * layout) requests and responses carry real payloads; handles are snapshot slots like notification handles and are released
 * with lbft_node_release_notification.  In reference mode (quirks bit 0 clear) the calls follow the reference simulator, where
 * a request is answered by the node that issued it (simulator.rs:446, quirk Q1): handles are payload-free tokens,
 * lbft_node_handle_request must be called on the requester itself (LBFT_ERR_UNSUPPORTED on any other node: a peer's answer
 * needs the payloads of quirks bit 0) and lbft_node_handle_response inserts nothing, exactly like data_sync.rs:209-240 over
 * records the node already holds.
 *   create_request  (data_sync.rs:66-71,179-181): the requester's epoch and known_quorum_certificate_rounds (record_store.rs:766-799)
 *   handle_request  (data_sync.rs:183-207) on the node it was sent to: the records the requester lacks (unknown_records,
 *                   record_store.rs:801-831), for every epoch from the requester's to the node's own -> a response handle
 *   handle_response (data_sync.rs:209-240) on the requester: inserts them epoch by epoch, processing commits in between */
int lbft_node_create_request(lbft_batch* b, size_t inst, uint32_t node, uint32_t* handle);
int lbft_node_handle_request(lbft_batch* b, size_t inst, uint32_t node, uint32_t request, uint32_t* response);
int lbft_node_handle_response(lbft_batch* b, size_t inst, uint32_t node, uint32_t peer, uint32_t response, int64_t node_time);
int lbft_node_view_get(lbft_batch* b, size_t inst, uint32_t node, lbft_node_view* out);

/* The same trait calls for many instances at once: ONE kernel launch and ONE synchronisation for the whole array instead of one per
 * call (a host driving thousands of simulators through ConsensusNode / DataSyncNode is otherwise launch-bound).  Every call of a
 * batch must address a different instance -- calls on one instance are ordered by the protocol and belong in successive batches
 * (LBFT_ERR_INVALID otherwise).  results[k] belongs to calls[k]; a call that found no free snapshot slot has results[k].status =
 * LBFT_ERR_FAULT (and the function returns LBFT_ERR_FAULT; the other results are valid).  Request / response calls need the
 * record-exchange layout (quirks bit 0): in reference mode they are host-side tokens (see above), use the single calls. */
#define LBFT_CALL_UPDATE_NODE 0          /* ConsensusNode::update_node(node_time) on `node`                    -> actions */
#define LBFT_CALL_CREATE_NOTIFICATION 1  /* DataSyncNode::create_notification on `node`                        -> handle */
#define LBFT_CALL_HANDLE_NOTIFICATION 2  /* handle_notification(handle) from `peer` on `node`                  -> should_sync */
#define LBFT_CALL_RELEASE_NOTIFICATION 3 /* drop one reference of `handle` (notification, request or response) */
#define LBFT_CALL_CREATE_REQUEST 4       /* DataSyncNode::create_request on `node`                              -> handle */
#define LBFT_CALL_HANDLE_REQUEST 5       /* handle_request(handle) on `node` (the peer that answers)            -> handle (the response) */
#define LBFT_CALL_HANDLE_RESPONSE 6      /* handle_response(handle, node_time) from `peer` on `node` */
typedef struct lbft_node_call {
  uint32_t op;       /* LBFT_CALL_* */
  uint32_t instance; /* at most one call per instance and batch */
  uint32_t node;
  uint32_t peer;
  uint32_t handle;
  uint32_t reserved;
  int64_t node_time;
} lbft_node_call;
typedef struct lbft_node_result {
  lbft_actions actions; /* LBFT_CALL_UPDATE_NODE */
  uint32_t handle;      /* create_notification / create_request / handle_request */
  uint32_t should_sync; /* handle_notification */
  int32_t status;       /* LBFT_OK or LBFT_ERR_FAULT */
  uint32_t reserved;
} lbft_node_result;
int lbft_node_calls(lbft_batch* b, const lbft_node_call* calls, size_t n, lbft_node_result* results);

/* Stand-alone device checks of the third-party arithmetic (tests): each runs a tiny kernel.
 *   leaders: out[r] = PacemakerState::leader(round r) (pacemaker.rs:100-109) for r < n_rounds
 *   delays:  n samples of RandomDelay (simulator.rs:110-118) from Xoshiro256**(seed)
 *   exp/log: lbft_math.h on the device, bit patterns in/out */
int lbft_device_leaders(int device, const uint64_t* voting_rights, uint32_t num_nodes, uint8_t* out, uint32_t n_rounds);
int lbft_device_sample_delays(int device, const lbft_config* cfg, uint64_t seed, int64_t* out, size_t n);
int lbft_device_exp_log(int device, const double* x, double* exp_out, double* log_out, size_t n);

const char* lbft_last_error(void);
/* "gfx950" build tag and ABI version. */
const char* lbft_build_info(void);

#ifdef __cplusplus
}
#endif
#endif /* LBFT_H */
