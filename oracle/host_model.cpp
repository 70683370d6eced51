// host_model.cpp -- TEST INFRASTRUCTURE: compiles the kernel logic (librabft_simulator_amd/csrc/lbft_core.h)
// for the host so that the compact structural model the HIP kernels execute can be differential-tested
// against the full-fidelity oracle on CPU-only machines (tests/test_host_model.py).  It is never linked
// into the product library and never used as a fallback.
#include <cmath>
#include <cstring>
#include <thread>
#include <type_traits>
#include <vector>

#include "../librabft_simulator_amd/csrc/lbft_core.h"
#include "../librabft_simulator_amd/csrc/lbft_save_node.h"
#include "../librabft_simulator_amd/csrc/lbft_tables.h"
#include "lbft_oracle.h"

using namespace lbft;

static const u64 ZX[257] = LBFT_ZIG_NORM_X_BITS_INIT;
static const u64 ZF[257] = LBFT_ZIG_NORM_F_BITS_INIT;
static const u64 ET[256] = LBFT_EXP_TAB_INIT;

// save_node images of instance 0's nodes after the next lbft_hostmodel_run_batch (test hook, not thread-safe): node k's image is written
// at image_buf + k * image_stride, its length (or (size_t)-1: unsupported) into image_lens[k]
static uint8_t* g_image_buf = nullptr;
static size_t g_image_stride = 0;
static size_t* g_image_lens = nullptr;
// ... and, when set, the result of a save -> scrub -> load -> save round trip of every node through lbft_save_node.h's loader
// (load_node_image, what lbft_batch_load_node runs): 0 = the second image equals the first byte for byte, 1 = it differs, < 0 = the loader's code
static int* g_roundtrip = nullptr;

extern "C" {

void lbft_hostmodel_capture_node_images(uint8_t* buf, size_t stride, size_t* lens) { g_image_buf = buf; g_image_stride = stride; g_image_lens = lens; }
void lbft_hostmodel_roundtrip_node_images(int* results) { g_roundtrip = results; }

typedef struct lbft_hostmodel_caps {
  uint32_t qcap, scap, bcap, lcap;
  uint32_t ql;  // queue slots held in the emulated LDS front (0 = HBM rows only)
  uint32_t qheap;  // 1 = binary-heap event queue (the device's large-network mode)
  uint32_t force_generic;  // 1 = run the step as the run-time-generic class SimT<K_GENERIC> instead of the specialised one
  uint32_t rcap;           // > 0: round-switch trace (DataWriter) with this many rounds per node
  uint32_t qcal;           // 1 = calendar event queue (needs max_clock <= LBFT_CAL_MAX_CLOCK and a class >= 1 kernel)
  uint32_t ring;           // > 0 (class 2 + calendar): the cooperative event loop (run_coop / coop_bulk, 64 emulated lanes) with a ring of this many pre-generated draws
  uint32_t ring_topup;     // draws the generator runs ahead per step
  uint32_t tw;             // tile width of the state layout (0 = 64; the device uses the lanes per wavefront outside kernel class 0)
  uint32_t keep_stores;    // lbft_batch_keep_retired_stores: the retired record stores are archived in full
} lbft_hostmodel_caps;

// Same outputs as lbft_oracle_run_batch, plus per-instance fault words and max queue/snapshot use.
int lbft_hostmodel_run_batch(const lbft_oracle_config* cfg, const lbft_hostmodel_caps* caps, const uint64_t* seeds,
                             size_t n_instances, int64_t max_clock, uint32_t threads, uint32_t* commit_counts,
                             uint64_t* active_rounds, uint64_t* last_states, lbft_oracle_commit* histories,
                             size_t history_cap, lbft_oracle_counters* counters, uint32_t* faults,
                             uint32_t* maxq_out, uint32_t* maxsnap_out, int64_t* round_switches /* [inst][rcap][n], INT64_MIN = none */,
                             uint32_t* max_rounds /* [inst] */,
                             uint64_t* record_hashes /* [inst][node][hash_cap][4]: SimT::committed_record_hashes, or NULL */, size_t hash_cap) {
  if ((cfg->quirks & ~3u) != 0 || cfg->num_nodes > LBFT_MAX_NODES) return -10;
  Params p;
  memset(&p, 0, sizeof(p));
  p.n = cfg->num_nodes;
  p.m = (u32)n_instances;
  p.stride = (u32)((n_instances + 63) / 64 * 64);
  p.tw = caps->tw ? caps->tw : 64;
  if (p.tw > 64 || (p.tw & (p.tw - 1))) return -13;
  p.rsh = 2;
  while ((1u << p.rsh) < 4u * p.tw) p.rsh++;
  p.qcap = caps->qcap; p.scap = caps->scap; p.bcap = caps->bcap; p.lcap = caps->lcap;
  p.max_clock = (i32)max_clock;
  p.ql = caps->ql;
  p.qheap = caps->qheap;
  p.rcap = caps->rcap;
  {  // as the device host code: epochs a node can go through are bounded by its commits
    u64 eauto = (u64)caps->bcap / (cfg->commands_per_epoch ? cfg->commands_per_epoch : 1) + 2;
    p.ecap = (u32)(eauto > 4096 ? 4096 : eauto);
    if (p.ecap < 64) p.ecap = 64;
  }
  p.qcal = caps->qcal;
  p.rarch_words = caps->keep_stores ? 1u : 0u;  // (compute_layout turns the flag into the entry size)
  p.ring = (caps->qcal && p.n > 32) ? caps->ring : 0; p.ring_topup = p.ring ? caps->ring_topup : 0;
  if (p.ring & (p.ring - 1)) return -12;
  if (p.qcal) { if (max_clock > LBFT_CAL_MAX_CLOCK || p.rcap) return -11; p.qheap = 1; p.ql = 0; }
  p.delay_model = cfg->delay_model;
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
  p.rot = cfg->rights_rotation % p.n;
  p.total_votes = 0;
  std::vector<u32> weights(p.n);
  for (u32 i = 0; i < p.n; i++) { weights[i] = cfg->voting_rights ? (u32)cfg->voting_rights[i] : 1; p.total_votes += weights[i]; }
  p.weights = weights.data();
  p.quorum = 2 * p.total_votes / 3 + 1;
  p.unit_weights = 1;
  for (u32 i = 0; i < p.n; i++) if (p.weights[i] != 1) p.unit_weights = 0;
  if (p.unit_weights) p.rot = 0;
  std::vector<i64> dur(4096);
  for (size_t k = 0; k < dur.size(); k++) dur[k] = f64_to_i64_sat((double)cfg->delta * std::pow((double)k, cfg->gamma));
  const u32 leader_len = 4096, leader_tables = p.rot ? p.n : 1;  // one table per shift of the rotating voting rights
  std::vector<u8> leaders((size_t)leader_len * leader_tables);
  for (u32 k = 0; k < leader_tables; k++)
    for (u32 r = 0; r < leader_len; r++) leaders[(size_t)k * leader_len + r] = (u8)compute_leader(p.weights, p.n, p.total_votes, r, k);
  p.dur_tab = dur.data(); p.dur_len = (u32)dur.size();
  p.leader_tab = leaders.data(); p.leader_len = leader_len;
  p.exp_tab = ET; p.zig_x = ZX; p.zig_f = ZF;
  compute_layout(p);
  std::vector<u32> state(state_words(p), 0);

  if (threads == 0) threads = 1;
  if (p.n > 32) p.qheap = 1;  // as the device host code does
  // the step runs as the size class the device would pick (SimT<0..2>); init and read-back use the generic class
  auto run_one = [&](auto& s, size_t i) {
    std::vector<u64> keys(p.ql ? p.ql : 1);
    std::vector<u32> metas(p.ql ? p.ql : 1);
    s.attach_queue(keys.data(), metas.data(), 1, p.ql);
    std::vector<u32> hcbr(32);  // the device's LDS copy of the hcbr buffers (class 0, n <= 4)
    if (p.ql) s.attach_hcbr(hcbr.data());
    std::vector<u32> window(32 * (1 + BC_WORDS), 0);  // the large-network kernels' LDS window of block records (32 entries, as the device's default)
    if (p.n > 32) s.attach_blk_window(window.data(), 32, 0);
    s.load_scalars();
    s.queue_to_lds();
    s.hcbr_to_lds();
    bool done;
    if constexpr (std::remove_reference<decltype(s)>::type::COOP) done = p.ring ? s.run_coop(true) : s.run();
    else done = s.run();
    s.queue_from_lds();
    s.hcbr_from_lds();
    s.store_scalars(done);
  };
  int cls = caps->force_generic ? 3 : sim_class(p);
  if (cls <= 2) {  // the tile width each class addresses at compile time (64-wide tiles or instance-major rows)
    p.tw = layout_tile_width(p);
    p.rsh = 2;
    while ((1u << p.rsh) < 4u * p.tw) p.rsh++;
  }
  auto worker = [&](u32 tid) {
    for (size_t i = tid; i < n_instances; i += threads) {
      { Sim s0(p, state.data(), (u32)i); s0.init(seeds[i]); }
      // emulate the device's launch structure: the LDS front of the queue is a cache of the HBM rows
      if (cls == K_SMALL && LBFT_C0_QUAD && sim_quad(p)) { SimT<K_HEADLINE> s(p, state.data(), (u32)i); run_one(s, i); }  // as the device dispatches (large batches)
      else if (cls == K_SMALL) { SimT<K_SMALL> s(p, state.data(), (u32)i); run_one(s, i); }
      else if (cls == K_MID && sim_lean1(p)) { SimT<K_MID_LEAN> s(p, state.data(), (u32)i); run_one(s, i); }  // as the device dispatches
      else if (cls == K_MID) { SimT<K_MID> s(p, state.data(), (u32)i); run_one(s, i); }
      else if (cls == K_LARGE && sim_lean_q1(p)) { SimT<K_LARGE_EXCHANGE> s(p, state.data(), (u32)i); run_one(s, i); }  // as the device dispatches
      else if (cls == K_LARGE && sim_lean(p)) { SimT<K_LARGE_LEAN> s(p, state.data(), (u32)i); run_one(s, i); }
      else if (cls == K_LARGE) { SimT<K_LARGE> s(p, state.data(), (u32)i); run_one(s, i); }
      else { Sim s(p, state.data(), (u32)i); run_one(s, i); }
    }
  };
  std::vector<std::thread> ts;
  for (u32 t = 1; t < threads; t++) ts.emplace_back(worker, t);
  worker(0);
  for (auto& t : ts) t.join();

  if (g_image_buf && g_image_lens) {  // ConsensusNode::save_node of every node of instance 0 (lbft_save_node.h, the code the product library runs)
    std::vector<u32> hw(p.total_words);
    for (u32 w = 0; w < p.total_words; w++) hw[w] = state[word_offset(p, 0, w)];
    for (u32 k = 0; k < p.n; k++) {
      std::vector<uint8_t> image;
      std::string err;
      int irc = build_node_image(p, hw.data(), k, weights.data(), cfg->delta, cfg->gamma, cfg->lambda, cfg->target_commit_interval, image, err);
      g_image_lens[k] = irc == 0 ? image.size() : (size_t)-1;
      if (irc == 0 && image.size() <= g_image_stride) memcpy(g_image_buf + (size_t)k * g_image_stride, image.data(), image.size());
      if (g_roundtrip && irc == 0) {
        // scrub everything of node k that is NodeState (its rows but the simulator's / context's words, its bit in every block's KNOWN / QC
        // set, its archive of retired stores), load the image, save again
        std::vector<u32> h2(hw);
        Params hp = p; hp.m = 1; hp.stride = 64; hp.tw = 1; hp.rsh = 2; hp.weights = weights.data();
        Sim s2(hp, h2.data(), 0);
        const u32 keep[] = {NF_STARTUP, NF_IGNORE_UNTIL, NF_LAST_TIMER_T, NF_TIMER_DUPS, NF_DUP_STAMP, NF_NEXT_CMD, NF_LAST_COMMITTED_BLK, NF_NCOMMITS};
        for (u32 f = 0; f < hp.node_words; f++) {
          bool kept = false;
          for (u32 q : keep) kept |= q == f;
          if (!kept) s2.nfms(k, f, 0xdeadbeefu);
        }
        for (u32 x = 1; x <= s2.ld(I_NBLOCKS); x++)
          for (u32 f : {(u32)B_KNOWN, (u32)B_QC}) {
            u32 w = k < 32 ? s2.bfw(x, f) : s2.bxw(x, f, k >> 5);
            s2.st(w, s2.ld(w) ^ (1u << (k & 31u)));  // (flipped: a loader that leaves bits alone is caught either way)
          }
        for (u32 q = 0; q < hp.ecap * hp.rarch_words; q++) s2.st(hp.off_rarch + k * hp.ecap * hp.rarch_words + q, 0x5a5a5a5au);
        // quirks bit 0: the snapshot-format summaries of the node's retired stores (what its peers' record exchange reads) -- zeroed here,
        // compared with the original rows after the load
        const u32 arch_words = (hp.quirks & 1u) ? s2.nfm(k, NF_EPOCH) * hp.snap_words : 0;
        const u32 arch0 = hp.off_arch + k * hp.ecap * hp.snap_words;
        const u32 my_epoch = hw[s2.nfw(k, NF_EPOCH)];
        const u32 arch_n = (hp.quirks & 1u) ? (my_epoch < hp.ecap ? my_epoch : hp.ecap) * hp.snap_words : 0;
        (void)arch_words;
        for (u32 q = 0; q < arch_n; q++) s2.st(arch0 + q, 0);
        std::string lerr;
        int lrc = load_node_image(p, h2.data(), k, weights.data(), cfg->delta, cfg->gamma, cfg->lambda, cfg->target_commit_interval, image.data(), image.size(),
                                  INT64_MAX, lerr);
        if (lrc != 0) g_roundtrip[k] = lrc;
        else {
          std::vector<uint8_t> again;
          int brc = build_node_image(p, h2.data(), k, weights.data(), cfg->delta, cfg->gamma, cfg->lambda, cfg->target_commit_interval, again, lerr);
          g_roundtrip[k] = brc != 0 ? brc - 100 : (again == image ? 0 : 1);
          for (u32 q = 0; q < arch_n && g_roundtrip[k] == 0; q++) {
            // (the request words of a response slot -- sqw -- are not part of a store summary; a skipped epoch id has no entry)
            u32 within = q % hp.snap_words;
            if (within >= S_FIXED_WORDS + 2 * hp.n + 2 * (hp.mw - 1)) continue;
            if (h2[arch0 + q] != hw[arch0 + q]) g_roundtrip[k] = 3;
          }
          // ... and the guard of node.rs:219-228: a node time before the image's own times is refused and nothing is written
          std::vector<u32> h3(h2);
          int grc = load_node_image(p, h3.data(), k, weights.data(), cfg->delta, cfg->gamma, cfg->lambda, cfg->target_commit_interval, image.data(), image.size(),
                                    INT64_MIN + 1, lerr);
          if (g_roundtrip[k] == 0 && (grc != -4 || h3 != h2)) g_roundtrip[k] = 2;
        }
      }
    }
    g_image_buf = nullptr; g_image_lens = nullptr; g_roundtrip = nullptr;
  }
  if (counters) memset(counters, 0, sizeof(*counters));
  int rc = 0;
  for (size_t i = 0; i < n_instances; i++) {
    Sim s(p, state.data(), (u32)i);
    s.load_scalars();
    if (faults) faults[i] = s.fault;
    if (s.fault) rc = 1;
    if (maxq_out) maxq_out[i] = s.maxq;
    if (maxsnap_out) maxsnap_out[i] = s.maxsnap;
    if (p.rcap && round_switches && max_rounds) {
      u32 mr = 0;
      for (u32 k = 0; k < p.n; k++) { u32 r = s.ld(p.off_trace + p.n * p.rcap + k); mr = r > mr ? r : mr; }
      max_rounds[i] = mr;
      for (u32 r = 0; r < p.rcap; r++)
        for (u32 k = 0; k < p.n; k++) {
          u32 t = s.ld(p.off_trace + k * p.rcap + r);
          round_switches[((size_t)i * p.rcap + r) * p.n + k] = t == 0xffffffffu ? INT64_MIN : (int64_t)(i32)t;
        }
    }
    if (record_hashes && hash_cap)
      for (u32 n = 0; n < p.n; n++) s.committed_record_hashes(n, record_hashes + ((size_t)i * p.n + n) * hash_cap * 4, (u32)hash_cap);
    u64 min_round = UINT64_MAX, min_commits = UINT64_MAX;
    for (u32 n = 0; n < p.n; n++) {
      size_t o = i * p.n + n;
      u32 nc = s.nfm(n, NF_NCOMMITS);
      u64 ar = s.nfm(n, NF_PM_ROUND);
      if (commit_counts) commit_counts[o] = nc;
      if (active_rounds) active_rounds[o] = ar;
      min_round = ar < min_round ? ar : min_round;
      min_commits = nc < min_commits ? nc : min_commits;
      Sip13 h;
      h.init();
      h.word(nc);
      for (u32 k = 0; k < nc; k++) {
        u32 b = s.ld(p.off_log + n * p.lcap + k);
        u64 proposer = s.blk_author(b), index = s.bf(b, B_CMD);
        i64 time = (i64)(i32)s.bf(b, B_TIME);
        h.word(proposer); h.word(index); h.word((u64)time);
        if (histories && k < history_cap) histories[o * history_cap + k] = lbft_oracle_commit{proposer, index, time};
      }
      if (last_states) last_states[o] = h.finish();
    }
    if (counters) {
      counters->events[0] += s.ev0; counters->events[1] += s.ev1; counters->events[2] += s.ev2; counters->events[3] += s.ev3;
      counters->rng_draws += s.rng.draws;
      counters->rounds += min_round; counters->commits += min_commits;
      counters->events_scheduled += s.stamp;
      if (s.maxq > counters->max_queue) counters->max_queue = s.maxq;
    }
  }
  return rc;
}

}  // extern "C"
