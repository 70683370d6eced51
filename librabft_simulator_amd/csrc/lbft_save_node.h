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
  { Sip13 h; h.init(); h.word(0); empty_state = h.finish(); }  // State of the empty ledger (epoch 0)
  // every block of the pool, in id order (a block's predecessor has a smaller id): hashes, States, QC hashes
  std::vector<BlockInfo> B(nblocks + 1);
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
    auto set_word_of = [&](u32 f, u32 k) { return k == 0 ? F(f) : s.ld(row0 + NF_FIXED_WORDS + 2 * n + s.am_idx(f) * (mw - 1) + k - 1); };
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
      for (u32 a : tc) put_timeout(htc, F(NF_FIXED_WORDS + tc_sel * n + a), a);
    } else w.b.push_back(0);
    {
      auto to = node_set(NF_TO_MASK);
      w.u64v(to.size());
      for (u32 a : to) { w.u64v(a); put_timeout(cur, F(NF_FIXED_WORDS + (1u - tc_sel) * n + a), a); }
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

}  // namespace lbft

#endif  // LBFT_SAVE_NODE_H
