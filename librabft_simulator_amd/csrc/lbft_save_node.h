// lbft_save_node.h -- ConsensusNode::save_node (librabft-v2/src/node.rs:233-238): the bincode image of one node's NodeState,
// rebuilt from ONE instance's state rows laid out contiguously on the host (tile width 1).  Host-only C++; used by the
// product library (lbft_hip.hip copies the instance's rows back from the GPU first) and by the CPU-only differential tests
// (oracle/host_model.cpp), which compare it byte for byte with the oracle's image (oracle/lbft_oracle.cpp lbft_oracle_save_node).
#ifndef LBFT_SAVE_NODE_H
#define LBFT_SAVE_NODE_H

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "lbft_core.h"

namespace lbft {

namespace save_node_detail {
struct BinOut {
  std::vector<uint8_t> b;
  // (little-endian host, like the image: words are appended with one copy each)
  void u64v(u64 v) { size_t o = b.size(); b.resize(o + 8); memcpy(b.data() + o, &v, 8); }
  void u32v(u32 v) { size_t o = b.size(); b.resize(o + 4); memcpy(b.data() + o, &v, 4); }
  void f64v(double d) { u64 u; memcpy(&u, &d, 8); u64v(u); }
  void opt(bool some, u64 v) { b.push_back(some ? 1 : 0); if (some) u64v(v); }
};
struct BlockInfo { u32 base; u32 round, prev, author, prev_round, pp, pp_round, epoch, cmd; i64 time; u64 hash, state, qc_hash; bool has_cs; u64 cs; std::vector<u32> voters; bool has_qc; };
}  // namespace save_node_detail


// Every block of the instance's pool, in id order (a block's predecessor has a smaller id): the hashes the reference gives the Block,
// the ledger State after it and its QuorumCertificate -- what identifies a record in a NodeState image.
inline void pool_block_infos(SimT<K_GENERIC>& s, const Params& hp, std::vector<save_node_detail::BlockInfo>& B, u64& empty_state) {
  using namespace save_node_detail;
  const u32 mw = hp.mw;
  const u32 nblocks = s.ld(I_NBLOCKS);
  { Sip13 h; h.init(); h.word(0); empty_state = h.finish(); }  // State of the empty ledger (epoch 0)
  B.assign(nblocks + 1, BlockInfo());
  for (u32 x = 1; x <= nblocks; x++) {
    BlockInfo& r = B[x];
    u32 link = s.bf(x, B_LINK);
    r.round = s.bf(x, B_ROUND); r.prev = link & 0xffffu; r.author = link >> 16; r.prev_round = s.bf(x, B_PREV_ROUND);
    r.pp = s.bf(x, B_PP) & 0xffffu; r.pp_round = s.bf(x, B_PP_ROUND); r.epoch = s.bf(x, B_EPOCH); r.cmd = s.bf(x, B_CMD);
    r.time = (i64)(i32)s.bf(x, B_TIME);
    // the ledger a block extends: its previous block, or -- first block of an epoch > 0 -- the block whose commit ended the previous
    // epoch (the proposer's initial state, record_store.rs:655-674): the proposer's commit of ledger depth - 1
    r.base = r.prev;
    if (!r.prev && r.epoch > 0) {
      u32 d = s.bf(x, B_DEPTH);
      r.base = d >= 2 ? s.ld(hp.off_log + r.author * hp.lcap + d - 2) : 0;
    }
    {  // State = DefaultHasher over the ledger history up to this block (simulated_context.rs:51-55,127-158)
      std::vector<u32> chain;
      for (u32 y = x; y; y = B[y].base) chain.push_back(y);
      Sip13 h; h.init(); h.word(chain.size());
      for (size_t k = chain.size(); k-- > 0;) { const BlockInfo& c = k == 0 ? r : B[chain[k]]; h.word(c.author); h.word(c.cmd); h.word((u64)c.time); }
      r.state = h.finish();
    }
    u64 prev_qc = r.prev ? B[r.prev].qc_hash : record_hash_epoch_id(r.epoch);
    r.hash = record_hash_block(r.author, r.cmd, r.time, prev_qc, r.round, r.author);
    r.has_cs = r.prev && r.pp && r.round == r.prev_round + 1 && r.prev_round == r.pp_round + 1;  // vote_committed_state (record_store.rs:237-255)
    r.cs = r.has_cs ? B[r.pp].state : 0;
    for (u32 w = 0; w < mw; w++)
      for (u32 m = w == 0 ? s.bf(x, B_VOTERS) : s.ld(s.bfw(x, B_WORDS + 3 * (mw - 1) + w - 1)); m; m &= m - 1) r.voters.push_back(32 * w + ctz32(m));
    r.has_qc = !r.voters.empty();
    r.qc_hash = 0;
    if (r.has_qc) {  // QuorumCertificate_ (record.rs:82-99)
      SipBytes hq; hq.init();
      const char name[] = "QuorumCertificate_::";
      for (u32 i = 0; i < sizeof(name) - 1; i++) hq.byte((u32)name[i]);
      hq.u64le(r.epoch); hq.u64le(r.round); hq.u64le(r.hash); hq.u64le(r.state); hq.option(r.has_cs, r.cs);
      hq.uleb(r.voters.size());
      for (u32 a : r.voters) { hq.u64le(a); hq.u64le(a); hq.u64le(record_hash_vote(r.epoch, r.round, r.hash, r.state, r.has_cs, r.cs, a)); }
      hq.u64le(r.author);
      r.qc_hash = hq.finish();
    }
  }
}

// `hw`: the instance's total_words rows; `dp`: the batch's parameters (layout); returns 0, or -3 (unsupported: the node has changed epoch)
inline int build_node_image(const Params& dp, u32* hw, u32 node, const u32* weights, i64 cfg_delta, double cfg_gamma, double cfg_lambda,
                            i64 cfg_tci, std::vector<uint8_t>& out, std::string& err) {
  using namespace save_node_detail;
  Params hp = dp;
  hp.m = 1; hp.stride = 64; hp.tw = 1; hp.rsh = 2; hp.weights = weights;
  Sim s(hp, hw, 0);
  const u32 n = hp.n, mw = hp.mw;
  const u32 epoch = s.nfm(node, NF_EPOCH);
  if (epoch != 0 && hp.rarch_words == 0) {
    err = "save_node: the node has changed epoch and the batch did not keep the retired record stores (past_record_stores, node.rs:43): "
          "call lbft_batch_keep_retired_stores before running it";
    return -3;
  }
  if (epoch > hp.ecap) { err = "save_node: more epochs than the archive of retired record stores holds"; return -3; }
  const u32 nblocks = s.ld(I_NBLOCKS);
  auto set_word = [&](u32 blk, u32 f, u32 k) { return k == 0 ? s.bf(blk, f) : s.ld(s.bxw(blk, f, k)); };
  auto in_set = [&](u32 blk, u32 f, u32 a) { return (set_word(blk, f, a >> 5) >> (a & 31u)) & 1u; };
  u64 empty_state;
  std::vector<BlockInfo> B;
  pool_block_infos(s, hp, B, empty_state);
  BinOut w;
  w.b.reserve(4096 + (size_t)nblocks * 400);
  auto put_vote = [&](u32 blk, u32 a) {  // Vote = SignedValue<Vote_> (record.rs:65-80)
    const BlockInfo& r = B[blk];
    w.u64v(r.epoch); w.u64v(r.round); w.u64v(r.hash); w.u64v(r.state); w.opt(r.has_cs, r.cs); w.u64v(a);
    w.u64v(a); w.u64v(record_hash_vote(r.epoch, r.round, r.hash, r.state, r.has_cs, r.cs, a));
  };
  // ---- RecordStoreState (record_store.rs:93-119) of epoch `e`: the node's current store (its rows) or one it retired (the verbatim
  // copy of its rows taken at that epoch change, SimT::retire_store) ----
  auto put_store = [&](u32 e, u32 row0) {  // row0: first word of the rows describing the store
    auto F = [&](u32 f) { return s.ld(row0 + f); };
    auto set_word_of = [&](u32 f, u32 k) { return k == 0 ? F(f) : s.ld(row0 + NF_FIXED_WORDS + s.am_idx(f) * (mw - 1) + k - 1); };
    auto node_set = [&](u32 f) { std::vector<u32> v; for (u32 k = 0; k < mw; k++) for (u32 m = set_word_of(f, k); m; m &= m - 1) v.push_back(32 * k + ctz32(m)); return v; };
    auto put_timeout = [&](u32 round, u32 hcbr, u32 a) {  // Timeout = SignedValue<Timeout_>
      w.u64v(e); w.u64v(round); w.u64v(hcbr); w.u64v(a); w.u64v(a); w.u64v(record_hash_timeout(e, round, hcbr, a));
    };
    const u64 initial_hash = record_hash_epoch_id(e);
    const u32 init_blk = F(NF_INIT_STATE_BLK);
    const u64 initial_state = init_blk ? B[init_blk].state : empty_state;
    w.u64v(e);
    const u32 shift = hp.rot ? (e * hp.rot) % n : 0;
    auto right = [&](u32 a) { u32 i = a + shift; return (u64)weights[i >= n ? i - n : i]; };
    w.u64v(n); for (u32 a = 0; a < n; a++) { w.u64v(a); w.u64v(right(a)); }   // configuration.authors
    w.u64v(n); for (u32 a = 0; a < n; a++) { w.u64v(a); w.u64v(right(a)); }   // configuration.voting_rights (ascending author)
    w.u64v(hp.total_votes);
    w.u64v(initial_hash); w.u64v(initial_state);
    {  // blocks / quorum_certificates: what the node's store of this epoch holds, ascending hash
      std::vector<std::pair<u64, u32>> bl, qc;
      for (u32 x = 1; x <= nblocks; x++) {
        if (B[x].epoch != e) continue;
        if (in_set(x, B_KNOWN, node)) bl.push_back({B[x].hash, x});
        if (in_set(x, B_QC, node)) qc.push_back({B[x].qc_hash, x});
      }
      std::sort(bl.begin(), bl.end()); std::sort(qc.begin(), qc.end());
      w.u64v(bl.size());
      for (auto& kv : bl) {
        const BlockInfo& r = B[kv.second];
        w.u64v(kv.first);
        w.u64v(r.author); w.u64v(r.cmd); w.u64v((u64)r.time); w.u64v(r.prev ? B[r.prev].qc_hash : initial_hash); w.u64v(r.round); w.u64v(r.author);
        w.u64v(r.author); w.u64v(r.hash);
      }
      w.u64v(qc.size());
      for (auto& kv : qc) {
        const BlockInfo& r = B[kv.second];
        w.u64v(kv.first);
        w.u64v(r.epoch); w.u64v(r.round); w.u64v(r.hash); w.u64v(r.state); w.opt(r.has_cs, r.cs);
        w.u64v(r.voters.size());
        for (u32 a : r.voters) { w.u64v(a); w.u64v(a); w.u64v(record_hash_vote(r.epoch, r.round, r.hash, r.state, r.has_cs, r.cs, a)); }
        w.u64v(r.author);
        w.u64v(r.author); w.u64v(r.qc_hash);
      }
    }
    const u32 pb = F(NF_PROPOSED_BLK), hqc = F(NF_HQC_BLK), hcc = F(NF_HCC_BLK);
    const u32 htc = F(NF_HTC_ROUND), cur = F(NF_CUR_ROUND), tc_sel = F(NF_TC_SEL);
    w.opt(pb != 0, pb ? B[pb].hash : 0);
    w.u64v(F(NF_HQC_ROUND)); w.u64v(hqc ? B[hqc].qc_hash : initial_hash);
    w.u64v(htc); w.u64v(cur); w.u64v(F(NF_HC_ROUND));
    w.opt(hcc != 0, hcc ? B[hcc].qc_hash : 0);
    if (htc) {  // highest_timeout_certificate: Option<Vec<Timeout>>
      auto tc = node_set(NF_TC_MASK);
      w.b.push_back(1); w.u64v(tc.size());
      for (u32 a : tc) put_timeout(htc, F(node_hcbr_off(mw) + tc_sel * n + a), a);
    } else w.b.push_back(0);
    {
      auto to = node_set(NF_TO_MASK);
      w.u64v(to.size());
      for (u32 a : to) { w.u64v(a); put_timeout(cur, F(node_hcbr_off(mw) + (1u - tc_sel) * n + a), a); }
    }
    {  // current_votes: HashMap<Author, Vote>, ascending author; the two ballot entries hold the voters by block
      auto v0 = node_set(NF_BAL0_AUTHORS), v1 = node_set(NF_BAL1_AUTHORS);
      const u32 b0 = F(NF_BAL0_BLK), b1 = F(NF_BAL1_BLK);
      std::vector<std::pair<u32, u32>> votes;
      for (u32 a : v0) votes.push_back({a, b0});
      for (u32 a : v1) votes.push_back({a, b1});
      std::sort(votes.begin(), votes.end());
      w.u64v(votes.size());
      for (auto& v : votes) { w.u64v(v.first); put_vote(v.second, v.first); }
      w.u64v(F(NF_TO_WEIGHT));
      const u32 el = F(NF_ELECTION);
      if ((el & 0xffu) == 0) {  // ElectionState::Ongoing { ballot }
        std::vector<std::pair<std::pair<u64, u64>, u64>> ballot;
        const u32 w0 = F(NF_BAL0_WEIGHT), w1 = F(NF_BAL1_WEIGHT);
        if (b0 && w0) ballot.push_back({{B[b0].hash, B[b0].state}, w0});
        if (b1 && w1) ballot.push_back({{B[b1].hash, B[b1].state}, w1});
        std::sort(ballot.begin(), ballot.end());
        w.u32v(0); w.u64v(ballot.size());
        for (auto& en : ballot) { w.u64v(en.first.first); w.u64v(en.first.second); w.u64v(en.second); }
      } else if ((el & 0xffu) == 1) {
        w.u32v(1); w.u64v(B[el >> 8].hash); w.u64v(B[el >> 8].state);
      } else w.u32v(2);
    }
  };
  put_store(epoch, s.nfw(node, 0));
  // ---- pacemaker: PacemakerState (pacemaker.rs:60-77) ----
  w.u64v(s.nfm(node, NF_PM_EPOCH)); w.u64v(s.nfm(node, NF_PM_ROUND));
  { u32 l = s.nfm(node, NF_PM_LEADER); w.opt(l != LBFT_NO_LEADER, l); }
  w.u64v((u64)(i64)(i32)s.nfm(node, NF_PM_START));
  w.u64v((u64)s.nfm(node, NF_PM_DUR_LO) | ((u64)s.nfm(node, NF_PM_DUR_HI) << 32));
  w.u64v((u64)cfg_delta); w.f64v(cfg_gamma); w.f64v(cfg_lambda);
  // ---- epoch_id, latest_voted_round, locked_round, latest_query_all_time, tracker ----
  w.u64v(epoch); w.u64v(s.nfm(node, NF_LVR)); w.u64v(s.nfm(node, NF_LOCKED)); w.u64v((u64)(i64)(i32)s.nfm(node, NF_LQAT));
  w.u64v(s.nfm(node, NF_TR_EPOCH)); w.u64v(s.nfm(node, NF_TR_HCR)); w.u64v((u64)(i64)(i32)s.nfm(node, NF_TR_LCT)); w.u64v((u64)cfg_tci);
  // ---- past_record_stores: HashMap<EpochId, RecordStoreState> (node.rs:43), ascending epoch.  A node retires exactly the stores of
  // the epochs it has been in: every epoch between two commits is entered (process_commits stops at an epoch change, node.rs:346), but
  // an epoch change may skip ids when commands_per_epoch is crossed more than once by ONE commit -- then no store of the skipped id
  // exists; a retired entry is recognisable by its epoch word.
  {
    std::vector<u32> past;
    for (u32 e = 0; e < epoch; e++) {
      u32 row0 = hp.off_rarch + (node * hp.ecap + e) * hp.rarch_words;
      bool used = e == 0 ? (s.ld(row0 + NF_CUR_ROUND) != 0) : (s.ld(row0 + NF_EPOCH) == e);  // (rows start zeroed; a live store has current_round >= 1)
      if (used) past.push_back(e);
    }
    w.u64v(past.size());
    for (u32 e : past) { w.u64v(e); put_store(e, hp.off_rarch + (node * hp.ecap + e) * hp.rarch_words); }
  }
  out.swap(w.b);
  return 0;
}


// ---- ConsensusNode::load_node (librabft-v2/src/node.rs:211-231): bincode::deserialize::<NodeState>(image) into the node's rows --------
// The inverse of build_node_image.  The event loop keeps structural ids where the reference keeps records, so every record the image
// names is looked up by its hash among the records of the instance's block pool (pool_block_infos): an image saved from this instance
// (any node, any earlier time), from another batch run with the same seed, or by the oracle / the reference for the same run loads;
// an image naming a record the pool does not hold is refused (-3) -- there is no block row to point at.  What is NodeState is written:
// the record store (the node's fixed words, timeout / vote sets and hcbr buffers, its KNOWN / QC bit in every block of the pool), the
// pacemaker, epoch, voting constraints, tracker, and the retired stores (past_record_stores: archive rows when the batch keeps them,
// the snapshot-format summaries of quirks bit 0).  What is NOT NodeState stays: the simulator's timer bookkeeping and startup time
// (SimulatedNode, simulator.rs:53-59) and the SmrContext (ledger, pending states: simulated_context.rs:75-83), which the reference
// passes to load_node separately.  `node_time`: the guard of node.rs:219-228 ("refusing to restore saved state from the future").
// Returns 0; -1 malformed image / another configuration; -3 unknown record / no room; -4 state from the future.  `hw` is only
// modified when 0 is returned.
namespace save_node_detail {
struct BinIn {
  const uint8_t* p; size_t n; size_t o; bool ok;
  u64 u64v() { if (o + 8 > n) { ok = false; o = n; return 0; } u64 v; memcpy(&v, p + o, 8); o += 8; return v; }
  u32 u32v() { if (o + 4 > n) { ok = false; o = n; return 0; } u32 v; memcpy(&v, p + o, 4); o += 4; return v; }
  double f64v() { u64 u = u64v(); double d; memcpy(&d, &u, 8); return d; }
  bool tag() { if (o + 1 > n) { ok = false; return false; } uint8_t t = p[o++]; if (t > 1) ok = false; return t == 1; }
  // a sequence length that the remaining bytes can hold (each element is at least `min_bytes`)
  u64 len(size_t min_bytes) { u64 k = u64v(); if (k > (n - o) / (min_bytes ? min_bytes : 1)) { ok = false; return 0; } return k; }
};
struct TimeoutImg { u64 round, hcbr, author; };
struct VoteImg { u64 round, block_hash, state, author; };
struct StoreImg {
  u64 epoch, total_votes, initial_hash, initial_state;
  std::vector<std::pair<u64, u64>> rights;
  std::vector<u64> blocks, qcs;  // keys (record hashes)
  bool has_pb; u64 pb;
  u64 hqc_round, hqc_hash, htc_round, cur_round, hc_round;
  bool has_hcc; u64 hcc;
  bool has_tc; std::vector<TimeoutImg> tc, to;
  std::vector<VoteImg> votes;
  u64 to_weight;
  u32 election;  // 0 ongoing, 1 won, 2 closed
  std::vector<std::pair<std::pair<u64, u64>, u64>> ballot;
  u64 won_hash, won_state;
};
inline void read_signature(BinIn& r) { r.u64v(); r.u64v(); }
inline TimeoutImg read_timeout(BinIn& r, u64& epoch) {
  TimeoutImg t; epoch = r.u64v(); t.round = r.u64v(); t.hcbr = r.u64v(); t.author = r.u64v(); read_signature(r); return t;
}
inline bool read_store(BinIn& r, StoreImg& st) {
  st.epoch = r.u64v();
  u64 na = r.len(16);
  for (u64 k = 0; k < na && r.ok; k++) { r.u64v(); r.u64v(); }  // configuration.authors (the same pairs follow as voting_rights)
  u64 nr = r.len(16);
  for (u64 k = 0; k < nr && r.ok; k++) { u64 a = r.u64v(), w = r.u64v(); st.rights.push_back({a, w}); }
  st.total_votes = r.u64v();
  st.initial_hash = r.u64v(); st.initial_state = r.u64v();
  u64 nb = r.len(72);
  for (u64 k = 0; k < nb && r.ok; k++) { st.blocks.push_back(r.u64v()); for (int q = 0; q < 6; q++) r.u64v(); read_signature(r); }
  u64 nq = r.len(73);
  for (u64 k = 0; k < nq && r.ok; k++) {
    st.qcs.push_back(r.u64v());
    r.u64v(); r.u64v(); r.u64v(); r.u64v(); if (r.tag()) r.u64v();
    u64 nv = r.len(24);
    for (u64 v = 0; v < nv && r.ok; v++) { r.u64v(); read_signature(r); }
    r.u64v(); read_signature(r);
  }
  st.has_pb = r.tag(); st.pb = st.has_pb ? r.u64v() : 0;
  st.hqc_round = r.u64v(); st.hqc_hash = r.u64v(); st.htc_round = r.u64v(); st.cur_round = r.u64v(); st.hc_round = r.u64v();
  st.has_hcc = r.tag(); st.hcc = st.has_hcc ? r.u64v() : 0;
  st.has_tc = r.tag();
  u64 e2 = 0;
  if (st.has_tc) { u64 k = r.len(48); for (u64 q = 0; q < k && r.ok; q++) { st.tc.push_back(read_timeout(r, e2)); if (e2 != st.epoch) r.ok = false; } }
  { u64 k = r.len(56); for (u64 q = 0; q < k && r.ok; q++) { u64 a = r.u64v(); TimeoutImg t = read_timeout(r, e2); if (e2 != st.epoch || t.author != a) r.ok = false; st.to.push_back(t); } }
  {
    u64 k = r.len(65);
    for (u64 q = 0; q < k && r.ok; q++) {
      u64 a = r.u64v();
      VoteImg v; u64 ve = r.u64v(); v.round = r.u64v(); v.block_hash = r.u64v(); v.state = r.u64v(); if (r.tag()) r.u64v(); v.author = r.u64v(); read_signature(r);
      if (ve != st.epoch || v.author != a) r.ok = false;
      st.votes.push_back(v);
    }
  }
  st.to_weight = r.u64v();
  st.election = r.u32v();
  st.won_hash = st.won_state = 0;
  if (st.election == 0) { u64 k = r.len(24); for (u64 q = 0; q < k && r.ok; q++) { u64 h = r.u64v(), sstate = r.u64v(), w = r.u64v(); st.ballot.push_back({{h, sstate}, w}); } }
  else if (st.election == 1) { st.won_hash = r.u64v(); st.won_state = r.u64v(); }
  else if (st.election != 2) r.ok = false;
  return r.ok;
}
}  // namespace save_node_detail

inline int load_node_image(const Params& dp, u32* hw, u32 node, const u32* weights, i64 cfg_delta, double cfg_gamma, double cfg_lambda,
                           i64 cfg_tci, const uint8_t* image, size_t image_len, i64 node_time, std::string& err) {
  using namespace save_node_detail;
  Params hp = dp;
  hp.m = 1; hp.stride = 64; hp.tw = 1; hp.rsh = 2; hp.weights = weights;
  const u32 n = hp.n, mw = hp.mw;
  // ---- 1. bincode::deserialize::<NodeState> ----
  BinIn r{image, image_len, 0, true};
  StoreImg cur;
  if (!read_store(r, cur)) { err = "load_node: malformed image (record store)"; return -1; }
  const u64 pm_epoch = r.u64v(), pm_round = r.u64v();
  const bool pm_has_leader = r.tag();
  const u64 pm_leader = pm_has_leader ? r.u64v() : 0;
  const i64 pm_start = (i64)r.u64v(), pm_dur = (i64)r.u64v(), im_delta = (i64)r.u64v();
  const double im_gamma = r.f64v(), im_lambda = r.f64v();
  const u64 epoch_id = r.u64v(), lvr = r.u64v(), locked = r.u64v();
  const i64 lqat = (i64)r.u64v();
  const u64 tr_epoch = r.u64v(), tr_hcr = r.u64v();
  const i64 tr_lct = (i64)r.u64v(), im_tci = (i64)r.u64v();
  std::vector<StoreImg> past;
  {
    u64 np = r.len(8);
    for (u64 k = 0; k < np && r.ok; k++) {
      u64 e = r.u64v();
      past.emplace_back();
      if (!read_store(r, past.back()) || past.back().epoch != e) r.ok = false;
    }
  }
  if (!r.ok || r.o != image_len) { err = "load_node: malformed image (not a bincode NodeState, or trailing bytes)"; return -1; }
  // ---- 2. the guard of node.rs:219-228 ----
  {
    i64 prev = lqat > tr_lct ? lqat : tr_lct;
    if (pm_start > prev) prev = pm_start;
    if (node_time < prev) { err = "load_node: refusing to restore saved state from the future"; return -4; }
  }
  // ---- 3. the image must describe a node of THIS configuration ----
  auto fits32 = [](u64 v) { return v < 0xffffffffull; };
  auto fits_time = [](i64 v) { return v >= -(i64)0x7fffffff && v <= (i64)0x7fffffff; };
  if (im_delta != cfg_delta || im_gamma != cfg_gamma || im_lambda != cfg_lambda || im_tci != cfg_tci) { err = "load_node: the image was saved under another NodeConfig (delta / gamma / lambda / target_commit_interval)"; return -1; }
  if (cur.epoch != epoch_id) { err = "load_node: record_store.epoch_id differs from NodeState.epoch_id"; return -1; }
  if (!fits32(epoch_id) || !fits32(pm_epoch) || !fits32(pm_round) || !fits32(lvr) || !fits32(locked) || !fits32(tr_epoch) || !fits32(tr_hcr) || !fits_time(pm_start) ||
      !fits_time(lqat) || !fits_time(tr_lct) || (pm_has_leader && pm_leader >= n)) { err = "load_node: a round / epoch / time of the image is outside the device's 32-bit fields"; return -1; }
  auto check_store = [&](const StoreImg& st) -> bool {
    if (st.rights.size() != n || st.total_votes != hp.total_votes || !fits32(st.epoch)) return false;
    const u32 shift = hp.rot ? (u32)((st.epoch * hp.rot) % n) : 0;
    for (u32 a = 0; a < n; a++) { u32 i = a + shift; if (st.rights[a].first != a || st.rights[a].second != (u64)weights[i >= n ? i - n : i]) return false; }
    if (st.initial_hash != record_hash_epoch_id(st.epoch)) return false;
    if (!fits32(st.hqc_round) || !fits32(st.htc_round) || !fits32(st.cur_round) || !fits32(st.hc_round) || !fits32(st.to_weight)) return false;
    for (auto& t : st.tc) if (t.author >= n || !fits32(t.hcbr) || t.round != st.htc_round) return false;
    for (auto& t : st.to) if (t.author >= n || !fits32(t.hcbr) || t.round != st.cur_round) return false;
    for (auto& v : st.votes) if (v.author >= n) return false;
    return true;
  };
  if (!check_store(cur)) { err = "load_node: the image's EpochConfiguration / record store does not belong to this batch's configuration"; return -1; }
  for (auto& st : past) {
    if (!check_store(st) || st.epoch >= epoch_id) { err = "load_node: a retired record store of the image does not belong to this batch's configuration"; return -1; }
    if (hp.rarch_words && st.epoch >= hp.ecap) { err = "load_node: more epochs than the archive of retired record stores holds"; return -3; }
  }
  // ---- 4. records by hash among the instance's block pool ----
  std::vector<u32> work(hw, hw + hp.total_words);  // (all writes go to a copy: nothing is modified unless the whole image loads)
  Sim s(hp, work.data(), 0);
  u64 empty_state;
  std::vector<BlockInfo> B;
  pool_block_infos(s, hp, B, empty_state);
  const u32 nblocks = (u32)B.size() - 1;
  auto find_block = [&](u64 h, u64 e) -> u32 { for (u32 x = 1; x <= nblocks; x++) if (B[x].epoch == e && B[x].hash == h) return x; return 0; };
  auto find_qc = [&](u64 h, u64 e) -> u32 { for (u32 x = 1; x <= nblocks; x++) if (B[x].epoch == e && B[x].has_qc && B[x].qc_hash == h) return x; return 0; };
  auto find_state = [&](u64 st) -> u32 { for (u32 x = 1; x <= nblocks; x++) if (B[x].state == st) return x; return 0; };
  bool unknown = false;
  auto put_set_bit = [&](u32 blk, u32 f, bool on) {
    u32 w = node < 32 ? s.bfw(blk, f) : s.bxw(blk, f, node >> 5);
    u32 v = s.ld(w), bit = 1u << (node & 31u);
    s.st(w, on ? (v | bit) : (v & ~bit));
  };
  // Writes the store `st` into the rows starting at `row0` (the node's own rows, or an archive entry) and its KNOWN / QC bits
  auto put_store = [&](const StoreImg& st, u32 row0, bool bits) {
    const u64 e = st.epoch;
    auto W = [&](u32 f, u32 v) { s.st(row0 + f, v); };
    auto set_word_at = [&](u32 f, u32 k) { return k == 0 ? row0 + f : row0 + NF_FIXED_WORDS + s.am_idx(f) * (mw - 1) + k - 1; };
    auto clear_set = [&](u32 f) { for (u32 k = 0; k < mw; k++) s.st(set_word_at(f, k), 0); };
    auto add_to_set = [&](u32 f, u32 a) { u32 w = set_word_at(f, a >> 5); s.st(w, s.ld(w) | (1u << (a & 31u))); };
    W(NF_EPOCH, (u32)e);
    u32 init_blk = 0;
    if (st.initial_state != empty_state) { init_blk = find_state(st.initial_state); if (!init_blk) unknown = true; }
    W(NF_INIT_STATE_BLK, init_blk);
    u32 pb = 0;
    if (st.has_pb) { pb = find_block(st.pb, e); if (!pb) unknown = true; }
    W(NF_PROPOSED_BLK, pb);
    W(NF_CUR_ROUND, (u32)st.cur_round);
    u32 hqc = 0;
    if (st.hqc_hash != st.initial_hash) { hqc = find_qc(st.hqc_hash, e); if (!hqc) unknown = true; }
    W(NF_HQC_ROUND, (u32)st.hqc_round); W(NF_HQC_BLK, hqc);
    W(NF_HTC_ROUND, (u32)st.htc_round); W(NF_HC_ROUND, (u32)st.hc_round);
    u32 hcc = 0;
    if (st.has_hcc) { hcc = find_qc(st.hcc, e); if (!hcc) unknown = true; }
    W(NF_HCC_BLK, hcc);
    // timeouts: the certificate in buffer 0, the current round's in buffer 1
    W(NF_TC_SEL, 0);
    clear_set(NF_TC_MASK); clear_set(NF_TO_MASK);
    for (u32 k = 0; k < 2 * n; k++) W(node_hcbr_off(mw) + k, 0);
    if (st.has_tc) for (auto& t : st.tc) { add_to_set(NF_TC_MASK, (u32)t.author); W(node_hcbr_off(mw) + (u32)t.author, (u32)t.hcbr); }
    for (auto& t : st.to) { add_to_set(NF_TO_MASK, (u32)t.author); W(node_hcbr_off(mw) + n + (u32)t.author, (u32)t.hcbr); }
    W(NF_TO_WEIGHT, (u32)st.to_weight);
    // votes: at most two distinct blocks (two ballot entries)
    clear_set(NF_BAL0_AUTHORS); clear_set(NF_BAL1_AUTHORS);
    u32 b0 = 0, b1 = 0; u64 w0 = 0, w1 = 0;
    const u32 shift = hp.rot ? (u32)((e * hp.rot) % n) : 0;
    auto right = [&](u32 a) { u32 i = a + shift; return (u64)weights[i >= n ? i - n : i]; };
    for (auto& v : st.votes) {
      u32 x = find_block(v.block_hash, e);
      if (!x) { unknown = true; continue; }
      if (b0 == 0 || b0 == x) { b0 = x; add_to_set(NF_BAL0_AUTHORS, (u32)v.author); w0 += right((u32)v.author); }
      else if (b1 == 0 || b1 == x) { b1 = x; add_to_set(NF_BAL1_AUTHORS, (u32)v.author); w1 += right((u32)v.author); }
      else unknown = true;  // (three ballot entries: only equivocating voters could produce them)
    }
    u32 el = st.election;
    if (st.election == 0) {  // Ongoing: the weights are the ballot's (a vote inserted while the election is not ongoing adds none)
      w0 = w1 = 0;
      for (auto& en : st.ballot) {
        u32 x = find_block(en.first.first, e);
        if (x && x == b0) w0 = en.second; else if (x && x == b1) w1 = en.second; else unknown = true;
      }
    } else if (st.election == 1) {
      u32 x = find_block(st.won_hash, e);
      if (!x) unknown = true;
      el = 1u | (x << 8);
    }
    W(NF_BAL0_BLK, b0); W(NF_BAL0_WEIGHT, (u32)w0); W(NF_BAL1_BLK, b1); W(NF_BAL1_WEIGHT, (u32)w1);
    W(NF_ELECTION, el);
    if (bits) {
      // the node's KNOWN / QC bit in every block of this epoch: exactly the records the image's store lists
      std::vector<uint8_t> known(nblocks + 1, 0), hasqc(nblocks + 1, 0);
      for (u64 h : st.blocks) { u32 x = find_block(h, e); if (!x) unknown = true; else known[x] = 1; }
      for (u64 h : st.qcs) { u32 x = find_qc(h, e); if (!x) unknown = true; else hasqc[x] = 1; }
      for (u32 x = 1; x <= nblocks; x++) if (B[x].epoch == e) { put_set_bit(x, B_KNOWN, known[x]); put_set_bit(x, B_QC, hasqc[x]); }
    }
    return hcc;
  };
  // The reference's load_node restores past_record_stores in full (node.rs:43).  A batch created without lbft_batch_keep_retired_stores has nowhere to
  // put them: loading such an image there would drop them silently and the node could not be saved again (save_node needs the archive) -- refused.
  if (!past.empty() && hp.rarch_words == 0) {
    err = "load_node: the image carries retired record stores (past_record_stores, node.rs:43) and this batch does not archive them: create it with "
          "lbft_batch_keep_retired_stores(batch, 1) before the run";
    return -3;
  }
  // every epoch the image has no store for: the node holds none of its records
  {
    std::vector<u64> have;
    have.push_back(cur.epoch);
    for (auto& st : past) have.push_back(st.epoch);
    for (u32 x = 1; x <= nblocks; x++)
      if (std::find(have.begin(), have.end(), (u64)B[x].epoch) == have.end()) { put_set_bit(x, B_KNOWN, false); put_set_bit(x, B_QC, false); }
  }
  put_store(cur, s.nfw(node, 0), true);
  // retired stores: archive rows (when the batch keeps them), the snapshot-format summaries of quirks bit 0, the previous store's
  // commit certificate (what notifications forward under quirks bit 1)
  u32 prev_hcc = 0; u64 prev_epoch = 0; bool any_prev = false;
  if (hp.rarch_words)
    for (u32 e = 0; e < hp.ecap && e < epoch_id; e++) {  // entries of epochs the image has no store for read as "no store"
      u32 row0 = hp.off_rarch + (node * hp.ecap + e) * hp.rarch_words;
      s.st(row0 + NF_EPOCH, 0); s.st(row0 + NF_CUR_ROUND, 0);
    }
  for (auto& st : past) {
    u32 hcc;
    if (hp.rarch_words) hcc = put_store(st, hp.off_rarch + (node * hp.ecap + (u32)st.epoch) * hp.rarch_words, true);
    else {  // (no archive: only the bits and the certificate are kept -- through a scratch copy of the node's row layout)
      std::vector<u32> keep(work.begin() + s.nfw(node, 0), work.begin() + s.nfw(node, 0) + hp.node_words);
      hcc = put_store(st, s.nfw(node, 0), true);
      std::copy(keep.begin(), keep.end(), work.begin() + s.nfw(node, 0));
    }
    if (!any_prev || st.epoch > prev_epoch) { prev_epoch = st.epoch; prev_hcc = hcc; any_prev = true; }
    if ((hp.quirks & 1u) && st.epoch < hp.ecap) {  // SimT::write_store_snapshot's words for this retired store (the record exchange reads them)
      const u32 base = hp.off_arch + (node * hp.ecap + (u32)st.epoch) * hp.snap_words;
      u32 hqc = st.hqc_hash != st.initial_hash ? find_qc(st.hqc_hash, st.epoch) : 0, pbk = st.has_pb ? find_block(st.pb, st.epoch) : 0;
      s.st(base + S_EPOCH, (u32)st.epoch); s.st(base + S_CERTS, hcc | (hqc << 16)); s.st(base + S_PROP_VOTE, pbk);
      s.st(base + S_TC_ROUND, (u32)st.htc_round); s.st(base + S_TO_ROUND, (u32)st.cur_round);
      std::vector<u32> tcm(mw, 0), tom(mw, 0);
      if (st.has_tc && st.htc_round) for (auto& t : st.tc) { tcm[t.author >> 5] |= 1u << (t.author & 31u); s.st(base + S_FIXED_WORDS + (u32)t.author, (u32)t.hcbr); }
      for (auto& t : st.to) { tom[t.author >> 5] |= 1u << (t.author & 31u); s.st(base + S_FIXED_WORDS + n + (u32)t.author, (u32)t.hcbr); }
      u32 xt = 0, xo = 0;
      for (u32 k = 0; k < mw; k++) {
        s.st(k == 0 ? base + S_TC_MASK : base + S_FIXED_WORDS + 2 * n + (k - 1), tcm[k]);
        s.st(k == 0 ? base + S_TO_MASK : base + S_FIXED_WORDS + 2 * n + (mw - 1) + (k - 1), tom[k]);
        if (k) { xt |= tcm[k]; xo |= tom[k]; }
      }
      if (mw > 1) {  // (LBFT_S_XFLAG: "an extension word of the set is non-zero", as SimT::write_store_snapshot marks it)
        s.st(base + S_TC_ROUND, (u32)st.htc_round | (xt ? LBFT_S_XFLAG : 0u)); s.st(base + S_TO_ROUND, (u32)st.cur_round | (xo ? LBFT_S_XFLAG : 0u));
      }
    }
  }
  if (unknown) {
    err = "load_node: the image names a record (block / quorum certificate / state) that this instance's block pool does not hold -- images load into "
          "the instance (or a same-seed run) they were saved from";
    return -3;
  }
  s.nfms(node, NF_PREV_EPOCH_HCC, prev_hcc);
  // ---- 5. pacemaker, voting constraints, tracker ----
  s.nfms(node, NF_PM_EPOCH, (u32)pm_epoch); s.nfms(node, NF_PM_ROUND, (u32)pm_round);
  s.nfms(node, NF_PM_LEADER, pm_has_leader ? (u32)pm_leader : LBFT_NO_LEADER);
  s.nfms(node, NF_PM_START, (u32)(i32)pm_start);
  s.nfms(node, NF_PM_DUR_LO, (u32)(u64)pm_dur); s.nfms(node, NF_PM_DUR_HI, (u32)((u64)pm_dur >> 32));
  s.nfms(node, NF_LVR, (u32)lvr); s.nfms(node, NF_LOCKED, (u32)locked); s.nfms(node, NF_LQAT, (u32)(i32)lqat);
  s.nfms(node, NF_TR_EPOCH, (u32)tr_epoch); s.nfms(node, NF_TR_HCR, (u32)tr_hcr); s.nfms(node, NF_TR_LCT, (u32)(i32)tr_lct);
  std::copy(work.begin(), work.end(), hw);
  return 0;
}

}  // namespace lbft

#endif  // LBFT_SAVE_NODE_H
