//! Rust binding of `liblbft_hip.so` (C ABI: `include/lbft.h`) for `novifinancial/librabft_simulator`.
//!
//! UNCOMPILED in the repository that ships it (no Rust toolchain there); written against the reference's
//! own types so that a maintainer can add it to the workspace as is.  Two layers:
//!
//! * [`GpuSimulator`] -- the call shape of `Simulator::new` + `Simulator::loop_until`
//!   (bft-lib/src/simulator.rs:200-250,380-475) for a vector of seeds;
//! * [`GpuNode`] -- one GPU-resident node behind the reference's traits, `ConsensusNode<Context>` and
//!   `DataSyncNode<Context>` exactly as bft-lib/src/interfaces.rs:37-86 declares them (lifetimes, `&mut Context`,
//!   `Async` / `AsyncResult` return types), so that `Simulator<GpuNode, SimulatedContext, ..>` or a
//!   `bft-driver` `CoreDriver` (bft-driver/src/core.rs:69,125-198) can own time and message delivery while
//!   the node state stays in HBM.
use std::os::raw::{c_char, c_int, c_void};
use std::sync::Arc;

use anyhow::{bail, ensure};
use bft_lib::{
    base_types::{Async, AsyncResult, NodeTime, Result},
    interfaces::{ConsensusNode, DataSyncNode, NodeUpdateActions},
    simulated_context::{Author, Command, SimulatedContext, State},
    simulator::GlobalTime,
    smr_context::Storage,
};
use futures::future;

// ------------------------------------------------------------------------------------------------
// C ABI (include/lbft.h)
// ------------------------------------------------------------------------------------------------
#[repr(C)]
pub struct LbftConfig {
    pub num_nodes: u32,
    pub delay_model: u32,
    pub mean: f64,
    pub variance: f64,
    pub uniform_lo: i64,
    pub uniform_hi: i64,
    pub commands_per_epoch: u64,
    pub target_commit_interval: i64,
    pub delta: i64,
    pub gamma: f64,
    pub lambda: f64,
    pub quirks: u32,
    pub equivocate_every: u32,
    pub voting_rights: *const u64,
    pub queue_capacity: u32,
    pub snapshot_capacity: u32,
    pub block_capacity: u32,
    pub log_capacity: u32,
    pub drop_per_million: u32,
    pub partition_size: u32,
    pub partition_start: i64,
    pub partition_end: i64,
    pub rights_rotation: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct LbftCommit {
    pub proposer: u64,
    pub index: u64,
    pub time: i64,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct LbftActions {
    pub next_scheduled_update: i64,
    pub should_send: [u64; 2],
    pub should_broadcast: u32,
    pub should_query_all: u32,
}

/// `lbft_node_call` / `lbft_node_result` (include/lbft.h): one entry of a batch of trait calls
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct LbftNodeCall {
    pub op: u32,
    pub instance: u32,
    pub node: u32,
    pub peer: u32,
    pub handle: u32,
    pub reserved: u32,
    pub node_time: i64,
}
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct LbftNodeResult {
    pub actions: LbftActions,
    pub handle: u32,
    pub should_sync: u32,
    pub status: i32,
    pub reserved: u32,
}
/// `lbft_counters` (include/lbft.h)
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct LbftCounters {
    pub events: [u64; 4],
    pub rng_draws: u64,
    pub rounds: u64,
    pub commits: u64,
    pub events_scheduled: u64,
    pub faulted_instances: u64,
    pub max_queue: u64,
    pub max_snapshots: u64,
    pub max_blocks: u64,
    pub launches: u64,
    pub timers_folded: u64,
    pub node_updates: u64,
}

pub const LBFT_OK: c_int = 0;
pub const LBFT_ERR_FAULT: c_int = -5;

extern "C" {
    fn lbft_last_error() -> *const c_char;
    fn lbft_batch_create(cfg: *const LbftConfig, seeds: *const u64, n: usize, device: c_int, out: *mut *mut c_void) -> c_int;
    fn lbft_batch_run_until(b: *mut c_void, max_clock: i64) -> c_int;
    fn lbft_batch_save_node(b: *const c_void, inst: usize, node: u32, buf: *mut c_void, cap: usize, len: *mut usize) -> c_int;
    /// node.rs:211-231 on the device: the bincode NodeState image into the GPU-resident node (guard "state from the future" included)
    fn lbft_batch_load_node(b: *mut c_void, inst: usize, node: u32, image: *const c_void, len: usize, node_time: i64) -> c_int;
    /// many trait calls (each on another instance) in one launch + one synchronisation: see include/lbft.h `lbft_node_calls`
    pub fn lbft_node_calls(b: *mut c_void, calls: *const LbftNodeCall, n: usize, results: *mut LbftNodeResult) -> c_int;
    /// the run's one collective, natively: ONE ncclAllGather of the 14 counter words per rank on the caller's ncclComm_t, reduced locally
    pub fn lbft_batch_counters_allgather_reduce(b: *mut c_void, nccl_comm: *mut c_void, out: *mut LbftCounters) -> c_int;
    /// past_record_stores (node.rs:43) kept in full on the device: save_node then also serves nodes that have changed epoch
    pub fn lbft_batch_keep_retired_stores(b: *mut c_void, enable: c_int) -> c_int;
    fn lbft_batch_commit_counts(b: *const c_void, out: *mut u32) -> c_int;
    fn lbft_batch_committed_history(b: *const c_void, inst: usize, node: u32, out: *mut LbftCommit, cap: usize, len: *mut usize) -> c_int;
    fn lbft_batch_last_committed_state(b: *const c_void, inst: usize, node: u32, out: *mut u64) -> c_int;
    fn lbft_batch_destroy(b: *mut c_void);
    // node-level interface
    fn lbft_batch_manual_begin(b: *mut c_void, max_clock: i64) -> c_int;
    fn lbft_node_update(b: *mut c_void, inst: usize, node: u32, node_time: i64, out: *mut LbftActions) -> c_int;
    fn lbft_node_create_notification(b: *mut c_void, inst: usize, node: u32, handle: *mut u32) -> c_int;
    fn lbft_node_handle_notification(b: *mut c_void, inst: usize, receiver: u32, sender: u32, handle: u32, should_sync: *mut u32) -> c_int;
    fn lbft_node_release_notification(b: *mut c_void, inst: usize, handle: u32) -> c_int;
    fn lbft_node_create_request(b: *mut c_void, inst: usize, node: u32, handle: *mut u32) -> c_int;
    fn lbft_node_handle_request(b: *mut c_void, inst: usize, node: u32, request: u32, response: *mut u32) -> c_int;
    fn lbft_node_handle_response(b: *mut c_void, inst: usize, node: u32, peer: u32, response: u32, node_time: i64) -> c_int;
}

fn check(rc: c_int) -> Result<()> {
    if rc == LBFT_OK {
        return Ok(());
    }
    let msg = unsafe { std::ffi::CStr::from_ptr(lbft_last_error()) }.to_string_lossy().into_owned();
    bail!("liblbft_hip: error {} ({})", rc, msg)
}

/// Owns one `lbft_batch` (all of its host and device memory).  Not `Sync`: one host thread per batch, like the
/// single-threaded reference (SURVEY.md 8b "Threading").
pub struct Batch {
    handle: *mut c_void,
    pub num_nodes: usize,
    pub num_instances: usize,
}
unsafe impl Send for Batch {}
impl Drop for Batch {
    fn drop(&mut self) {
        unsafe { lbft_batch_destroy(self.handle) }
    }
}

fn make_config(num_nodes: usize, mean: f64, variance: f64, commands_per_epoch: usize, config: &librabft_v2::node::NodeConfig, quirks: u32) -> LbftConfig {
    LbftConfig {
        num_nodes: num_nodes as u32,
        delay_model: 0, // RandomDelay::new(mean, variance): LogNormal (simulator.rs:99-106)
        mean,
        variance,
        uniform_lo: 0,
        uniform_hi: 0,
        commands_per_epoch: commands_per_epoch as u64,
        target_commit_interval: config.target_commit_interval.0,
        delta: config.delta.0,
        gamma: config.gamma,
        lambda: config.lambda,
        quirks,
        equivocate_every: 0,
        voting_rights: std::ptr::null(), // SimulatedContext: every author has weight 1 (simulated_context.rs:209-216)
        queue_capacity: 0,
        snapshot_capacity: 0,
        block_capacity: 0,
        log_capacity: 0,
        drop_per_million: 0,
        partition_size: 0,
        partition_start: 0,
        partition_end: 0,
        rights_rotation: 0,
    }
}

// ------------------------------------------------------------------------------------------------
// Batch level: Simulator::new + loop_until for many seeds
// ------------------------------------------------------------------------------------------------
/// `Simulator::new(rng_seed, num_nodes, RandomDelay::new(mean, variance), context_factory)` for every seed of
/// `rng_seeds`; the `context_factory` closure of librabft-v2/src/main.rs:23-34 only carries
/// `commands_per_epoch` and the `NodeConfig`, which travel in `LbftConfig`.
pub struct GpuSimulator {
    batch: Batch,
}

impl GpuSimulator {
    pub fn new(rng_seeds: &[u64], num_nodes: usize, mean: f64, variance: f64, commands_per_epoch: usize,
               config: &librabft_v2::node::NodeConfig) -> Result<Self> {
        let cfg = make_config(num_nodes, mean, variance, commands_per_epoch, config, 0);
        let mut handle = std::ptr::null_mut();
        check(unsafe { lbft_batch_create(&cfg, rng_seeds.as_ptr(), rng_seeds.len(), 0, &mut handle) })?;
        Ok(GpuSimulator { batch: Batch { handle, num_nodes, num_instances: rng_seeds.len() } })
    }

    /// `loop_until(GlobalTime(max_clock), None)`: per instance and node what `context.committed_history()` and
    /// `context.last_committed_state()` return in the reference (simulated_context.rs:98-100,194-196).
    pub fn loop_until(&mut self, max_clock: GlobalTime) -> Result<Vec<Vec<(Vec<(Command, NodeTime)>, State)>>> {
        check(unsafe { lbft_batch_run_until(self.batch.handle, max_clock.0) })?;
        let mut counts = vec![0u32; self.batch.num_instances * self.batch.num_nodes];
        check(unsafe { lbft_batch_commit_counts(self.batch.handle, counts.as_mut_ptr()) })?;
        let mut out = Vec::with_capacity(self.batch.num_instances);
        for i in 0..self.batch.num_instances {
            let mut per_node = Vec::with_capacity(self.batch.num_nodes);
            for n in 0..self.batch.num_nodes {
                let mut len = counts[i * self.batch.num_nodes + n] as usize;
                let mut buf = vec![LbftCommit::default(); len.max(1)];
                check(unsafe { lbft_batch_committed_history(self.batch.handle, i, n as u32, buf.as_mut_ptr(), len, &mut len) })?;
                let mut state = 0u64;
                check(unsafe { lbft_batch_last_committed_state(self.batch.handle, i, n as u32, &mut state) })?;
                let history = buf[..len]
                    .iter()
                    .map(|c| (Command { proposer: Author(c.proposer as usize), index: c.index as usize }, NodeTime(c.time)))
                    .collect();
                per_node.push((history, State(state)));
            }
            out.push(per_node);
        }
        Ok(out)
    }
}

// ------------------------------------------------------------------------------------------------
// Node level: the reference's traits over GPU-resident node state
// ------------------------------------------------------------------------------------------------
/// A batch in node-level mode (`lbft_batch_manual_begin`: initial node states, no event loop), shared by the
/// `GpuNode`s of its instances.
pub struct NodeBatch {
    batch: Batch,
}

impl NodeBatch {
    /// `quirks`: 0 = the reference simulator's routing (a request is answered by its requester, simulator.rs:446);
    /// 1 / 3 = requests answered by the peer with real payloads, as `bft-driver` routes them (core.rs:174-178).
    pub fn new(rng_seeds: &[u64], num_nodes: usize, commands_per_epoch: usize, config: &librabft_v2::node::NodeConfig,
               quirks: u32, max_clock: i64) -> Result<Arc<Self>> {
        let cfg = make_config(num_nodes, 10.0, 4.0, commands_per_epoch, config, quirks);
        let mut handle = std::ptr::null_mut();
        check(unsafe { lbft_batch_create(&cfg, rng_seeds.as_ptr(), rng_seeds.len(), 0, &mut handle) })?;
        let batch = Batch { handle, num_nodes, num_instances: rng_seeds.len() };
        // ConsensusNode::save_node serialises past_record_stores (node.rs:43): a node-level session keeps the retired stores in full
        check(unsafe { lbft_batch_keep_retired_stores(batch.handle, 1) })?;
        check(unsafe { lbft_batch_manual_begin(batch.handle, max_clock) })?;
        Ok(Arc::new(NodeBatch { batch }))
    }

    pub fn node(self: &Arc<Self>, instance: usize, author: Author) -> GpuNode {
        GpuNode { batch: self.clone(), inst: instance, author: author.0 as u32, last_saved: NodeTime(std::i64::MIN) }
    }
}

std::thread_local! {
    /// `ConsensusNode::load_node(context, clock)` is a constructor without `self`: the GPU-resident state it attaches to
    /// is looked up here (the role of the context's storage in the reference, node.rs:211-231).
    static ATTACH: std::cell::RefCell<Option<(Arc<NodeBatch>, usize)>> = std::cell::RefCell::new(None);
}
/// Selects the batch instance that the next `GpuNode::load_node` calls attach to.
pub fn attach_loads_to(batch: &Arc<NodeBatch>, instance: usize) {
    ATTACH.with(|a| *a.borrow_mut() = Some((batch.clone(), instance)));
}

/// One GPU-resident LibraBFTv2 node (`NodeState`, librabft-v2/src/node.rs:28-45) behind the reference's traits.
pub struct GpuNode {
    batch: Arc<NodeBatch>,
    inst: usize,
    author: u32,
    last_saved: NodeTime,
}
/// A device-side message: the handle of a snapshot slot (or, for requests / responses in reference mode, a payload-free
/// token).  The simulator clones a notification once per receiver (simulator.rs:348-354): clones share the slot, which is
/// released when the last clone is dropped.
pub struct DeviceMessage {
    batch: Arc<NodeBatch>,
    inst: usize,
    pub sender: u32,
    pub handle: u32,
}
impl Drop for DeviceMessage {
    fn drop(&mut self) {
        let _ = unsafe { lbft_node_release_notification(self.batch.batch.handle, self.inst, self.handle) };
    }
}
/// `DataSyncNode::Notification` / `Request` / `Response`
#[derive(Clone)]
pub struct GpuNotification(pub Arc<DeviceMessage>);
#[derive(Clone)]
pub struct GpuRequest(pub Arc<DeviceMessage>);
#[derive(Clone)]
pub struct GpuResponse(pub Arc<DeviceMessage>);

const SAVE_KEY: &str = "lbft_hip_node_saved_at";
/// The key under which the reference stores `bincode::serialize(&NodeState)` (librabft-v2/src/node.rs:233-238).
const NODE_IMAGE_KEY: &str = "node_state";

impl GpuNode {
    fn message(&self, handle: u32) -> Arc<DeviceMessage> {
        Arc::new(DeviceMessage { batch: self.batch.clone(), inst: self.inst, sender: self.author, handle })
    }
}

impl ConsensusNode<SimulatedContext> for GpuNode {
    /// node.rs:211-231: `bincode::deserialize(read_value("node_state"))` + the guard "refusing to restore saved state from the
    /// future".  The image the context holds under the reference's key -- written by `save_node` below, or by a reference
    /// `NodeState` of the same run -- is loaded INTO the GPU-resident node (`lbft_batch_load_node`: records are found by their
    /// hashes in the instance's block pool, the guard runs in the library against `clock`); without an image the node attaches
    /// to the state resident in HBM (a fresh batch: `NodeState::make_initial_state`), exactly as the reference's simulator falls
    /// back to `make_initial_state` when `load_node` finds no value (simulator.rs:221).
    fn load_node(context: &mut SimulatedContext, clock: NodeTime) -> AsyncResult<Self> {
        let attached = ATTACH.with(|a| a.borrow().clone());
        let author = context.author();
        Box::pin(async move {
            let (batch, inst) = match attached {
                Some(x) => x,
                None => bail!("attach_loads_to(batch, instance) first"),
            };
            let mut node = batch.node(inst, author);
            if let Some(image) = context.read_value(NODE_IMAGE_KEY.to_string()).await? {
                ensure!(!image.is_empty(), "the last save_node of this node failed: no image to restore");
                let rc = unsafe {
                    lbft_batch_load_node(batch.batch.handle, inst, author.0 as u32, image.as_ptr() as *const c_void, image.len(), clock.0)
                };
                check(rc)?;  // LBFT_ERR_STATE = "refusing to restore saved state from the future" (lbft_last_error has the text)
            }
            if let Some(bytes) = context.read_value(SAVE_KEY.to_string()).await? {
                ensure!(bytes.len() == 8, "corrupt save marker");
                let mut b = [0u8; 8];
                b.copy_from_slice(&bytes);
                node.last_saved = NodeTime(i64::from_le_bytes(b));
                ensure!(node.last_saved <= clock, "refusing to restore saved state from the future");
            }
            Ok(node)
        })
    }

    /// node.rs:240-304 on the device (`lbft_node_update`).
    fn update_node(&mut self, _context: &mut SimulatedContext, clock: NodeTime) -> NodeUpdateActions<SimulatedContext> {
        let mut a = LbftActions::default();
        check(unsafe { lbft_node_update(self.batch.batch.handle, self.inst, self.author, clock.0, &mut a) })
            .expect("lbft_node_update");
        self.last_saved = clock;
        NodeUpdateActions {
            next_scheduled_update: NodeTime(a.next_scheduled_update),
            should_send: (0..128usize)
                .filter(|i| (a.should_send[i / 64] >> (i % 64)) & 1 == 1)
                .map(Author)
                .collect(),
            should_broadcast: a.should_broadcast != 0,
            should_query_all: a.should_query_all != 0,
        }
    }

    /// node.rs:233-238: `bincode::serialize(&NodeState)` under the reference's key.  `lbft_batch_save_node` builds that image
    /// from the device state (HashMaps in ascending key order; the reference's `load_node` accepts any order), so a
    /// reference node can be restored from what a GPU node saved.  Nodes that have changed epoch need the batch to keep the
    /// retired record stores (`lbft_batch_keep_retired_stores`, called by `GpuBatch::new` before the session starts); if the
    /// image cannot be built the call FAILS -- a stale image of an earlier save is never left under the reference's key
    /// (round-2 advisor) and the save marker does not advance.
    fn save_node<'a>(&'a mut self, context: &'a mut SimulatedContext) -> AsyncResult<'a, ()> {
        let marker = self.last_saved.0.to_le_bytes().to_vec();
        let handle = self.batch.batch.handle as *const c_void;
        let (inst, author) = (self.inst, self.author);
        let mut len = 0usize;
        let mut image = Vec::new();
        let mut rc = unsafe { lbft_batch_save_node(handle, inst, author, std::ptr::null_mut(), 0, &mut len) };
        if rc == 0 {
            image.resize(len, 0u8);
            rc = unsafe { lbft_batch_save_node(handle, inst, author, image.as_mut_ptr() as *mut c_void, len, &mut len) };
        }
        Box::pin(async move {
            if rc != 0 {
                // overwrite whatever an earlier save left: an empty value does not deserialise as a NodeState, so load_node fails loudly
                context.store_value(NODE_IMAGE_KEY.to_string(), Vec::new()).await?;
                return check(rc);
            }
            context.store_value(NODE_IMAGE_KEY.to_string(), image).await?;
            context.store_value(SAVE_KEY.to_string(), marker).await
        })
    }
}

impl DataSyncNode<SimulatedContext> for GpuNode {
    type Notification = GpuNotification;
    type Request = GpuRequest;
    type Response = GpuResponse;

    /// data_sync.rs:82-111
    fn create_notification(&self, _context: &SimulatedContext) -> GpuNotification {
        let mut h = 0u32;
        check(unsafe { lbft_node_create_notification(self.batch.batch.handle, self.inst, self.author, &mut h) })
            .expect("lbft_node_create_notification");
        GpuNotification(self.message(h))
    }

    /// data_sync.rs:66-71,179-181
    fn create_request(&self, _context: &SimulatedContext) -> GpuRequest {
        let mut h = 0u32;
        check(unsafe { lbft_node_create_request(self.batch.batch.handle, self.inst, self.author, &mut h) })
            .expect("lbft_node_create_request");
        GpuRequest(self.message(h))
    }

    /// data_sync.rs:183-207.  In reference mode (quirks bit 0 clear) only the requester itself can answer
    /// (simulator.rs:446); with quirks bit 0 any peer can (bft-driver/src/core.rs:174-178).
    fn handle_request<'a>(&'a self, _context: &'a mut SimulatedContext, request: GpuRequest) -> Async<'a, GpuResponse> {
        let mut h = 0u32;
        check(unsafe { lbft_node_handle_request(self.batch.batch.handle, self.inst, self.author, request.0.handle, &mut h) })
            .expect("lbft_node_handle_request");
        Box::pin(future::ready(GpuResponse(self.message(h))))
    }

    /// data_sync.rs:113-177; `Some(request)` when the reference's `should_sync` is set.
    fn handle_notification<'a>(&'a mut self, context: &'a mut SimulatedContext, notification: GpuNotification)
        -> Async<'a, Option<GpuRequest>> {
        let mut sync = 0u32;
        check(unsafe {
            lbft_node_handle_notification(self.batch.batch.handle, self.inst, self.author, notification.0.sender, notification.0.handle, &mut sync)
        })
        .expect("lbft_node_handle_notification");
        let request = if sync != 0 { Some(self.create_request(context)) } else { None };
        Box::pin(future::ready(request))
    }

    /// data_sync.rs:209-240
    fn handle_response<'a>(&'a mut self, _context: &'a mut SimulatedContext, response: GpuResponse, clock: NodeTime) -> Async<'a, ()> {
        check(unsafe {
            lbft_node_handle_response(self.batch.batch.handle, self.inst, self.author, response.0.sender, response.0.handle, clock.0)
        })
        .expect("lbft_node_handle_response");
        Box::pin(future::ready(()))
    }
}
