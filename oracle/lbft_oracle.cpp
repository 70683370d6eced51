// lbft_oracle.cpp -- CPU oracle: a faithful restatement of the reference hot path.
//
// TEST INFRASTRUCTURE ONLY (see lbft_oracle.h).  Every function cites the reference file:line it
// follows (paths relative to the reference checkout; `simulator.rs`, `simulated_context.rs`,
// `configuration.rs`, `base_types.rs`, `smr_context.rs` live in bft-lib/src/, the rest in
// librabft-v2/src/).  Third-party arithmetic that is NOT under the reference tree (no Cargo.lock;
// bft-lib/Cargo.toml:14-27) is restated from the published algorithms of rand 0.8, rand_xoshiro 0.6,
// rand_distr 0.4, Rust std SipHash-1-3 (DefaultHasher, keys 0/0), bcs 0.1 and serde-name 0.1.
//
// Parity pin: both golden integration tests of the reference (librabft-v2/tests/simulated_run.rs)
// are reproduced bit-exactly, including State(..) values, see tests/test_oracle_golden.py.
//
// Deliberate canonicalisation: Rust HashMap iteration order is random per process;
// wherever the reference iterates a HashMap<Author, _> (TC member order record_store.rs:532-534, QC
// vote order :709-719, timeouts() :749-756) this file iterates in ascending author order.
#include "lbft_oracle.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <optional>
#include <queue>
#include <set>
#include <string>
#include <thread>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../librabft_simulator_amd/csrc/lbft_math.h"    // strict math mode only (math_mode == 1)
#include "../librabft_simulator_amd/csrc/lbft_tables.h"  // generated data tables

namespace {

using u8 = uint8_t;
using u32 = uint32_t;
using u64 = uint64_t;
using i64 = int64_t;
typedef unsigned __int128 u128;

struct Panic {
  std::string msg;
};
[[noreturn]] void panic(const std::string& m) { throw Panic{m}; }

// ------------------------------------------------------------------------------------------------
// SipHash-1-3 with keys (0,0) == Rust std `DefaultHasher::new()` (pacemaker.rs:101-108,
// simulated_context.rs:51-55,238-242).
// ------------------------------------------------------------------------------------------------
inline u64 rotl(u64 x, int b) { return (x << b) | (x >> (64 - b)); }

u64 siphash13(const u8* data, size_t len) {
  u64 v0 = 0x736f6d6570736575ULL, v1 = 0x646f72616e646f6dULL, v2 = 0x6c7967656e657261ULL,
      v3 = 0x7465646279746573ULL;
  auto round = [&]() {
    v0 += v1; v1 = rotl(v1, 13); v1 ^= v0; v0 = rotl(v0, 32);
    v2 += v3; v3 = rotl(v3, 16); v3 ^= v2;
    v0 += v3; v3 = rotl(v3, 21); v3 ^= v0;
    v2 += v1; v1 = rotl(v1, 17); v1 ^= v2; v2 = rotl(v2, 32);
  };
  size_t i = 0;
  for (; i + 8 <= len; i += 8) {
    u64 m;
    memcpy(&m, data + i, 8);  // little-endian host
    v3 ^= m; round(); v0 ^= m;
  }
  u64 b = (u64)len << 56;
  for (size_t j = 0; i + j < len; j++) b |= (u64)data[i + j] << (8 * j);
  v3 ^= b; round(); v0 ^= b;
  v2 ^= 0xff;
  round(); round(); round();
  return v0 ^ v1 ^ v2 ^ v3;
}

struct Bytes {
  std::vector<u8> b;
  void raw(const char* s) { b.insert(b.end(), s, s + strlen(s)); }
  void u64le(u64 v) { for (int i = 0; i < 8; i++) b.push_back((u8)(v >> (8 * i))); }
  void byte(u8 v) { b.push_back(v); }
  void uleb(u64 v) {  // BCS sequence length
    while (v >= 0x80) { b.push_back((u8)(v | 0x80)); v >>= 7; }
    b.push_back((u8)v);
  }
  u64 hash() const { return siphash13(b.data(), b.size()); }
};

// ------------------------------------------------------------------------------------------------
// rand_xoshiro 0.6 Xoshiro256StarStar (+ SplitMix64 seeding), rand 0.8 gen_range / shuffle,
// rand_distr 0.4 StandardNormal (ziggurat) and LogNormal.
// ------------------------------------------------------------------------------------------------
struct Xoshiro {
  u64 s[4];
  u64 draws = 0;
  explicit Xoshiro(u64 seed) {  // Xoshiro256StarStar::seed_from_u64 (simulator.rs:212, configuration.rs:66)
    u64 x = seed;
    for (int i = 0; i < 4; i++) {
      x += 0x9e3779b97f4a7c15ULL;
      u64 z = x;
      z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
      z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
      s[i] = z ^ (z >> 31);
    }
  }
  u64 next_u64() {
    draws++;
    u64 r = rotl(s[1] * 5, 7) * 9;
    u64 t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl(s[3], 45);
    return r;
  }
  u32 next_u32() { return (u32)(next_u64() >> 32); }
  // rand 0.8 UniformInt::<u64/usize>::sample_single(0, n)
  u64 gen_range_u64(u64 n) {
    u64 zone = (n << __builtin_clzll(n)) - 1;
    for (;;) {
      u64 v = next_u64();
      u128 m = (u128)v * n;
      if ((u64)m <= zone) return (u64)(m >> 64);
    }
  }
  // rand 0.8 UniformInt::<u32>::sample_single(0, n)
  u32 gen_range_u32(u32 n) {
    u32 zone = (n << __builtin_clz(n)) - 1;
    for (;;) {
      u32 v = next_u32();
      u64 m = (u64)v * n;
      if ((u32)m <= zone) return (u32)(m >> 32);
    }
  }
  // rand 0.8 SliceRandom::shuffle (simulator.rs:343,370)
  template <class T>
  void shuffle(std::vector<T>& v) {
    for (size_t i = v.size(); i-- > 1;) {
      size_t j = gen_range_u32((u32)(i + 1));
      std::swap(v[i], v[j]);
    }
  }
};

const u64 ZIG_X_BITS[257] = LBFT_ZIG_NORM_X_BITS_INIT;
const u64 ZIG_F_BITS[257] = LBFT_ZIG_NORM_F_BITS_INIT;
const u64 EXP_TAB[256] = LBFT_EXP_TAB_INIT;
inline double zig_x(int i) { return lbft_asdouble(ZIG_X_BITS[i]); }
inline double zig_f(int i) { return lbft_asdouble(ZIG_F_BITS[i]); }

struct Math {
  int mode;  // 0 host libm, 1 lbft_math.h
  double exp(double x) const { return mode ? lbft_exp(x, EXP_TAB) : std::exp(x); }
  double log(double x) const { return mode ? lbft_log(x) : std::log(x); }
};

// rand_distr 0.4 StandardNormal::sample / utils::ziggurat
double standard_normal(Xoshiro& rng, const Math& m) {
  const double R = lbft_asdouble(LBFT_ZIG_NORM_R_BITS);
  for (;;) {
    u64 bits = rng.next_u64();
    int i = (int)(bits & 0xff);
    double u = lbft_asdouble((1024ULL << 52) | (bits >> 12)) - 3.0;  // into_float_with_exponent(1) - 3.0
    double x = u * zig_x(i);
    if (std::fabs(x) < zig_x(i + 1)) return x;
    if (i == 0) {  // zero_case: sample from the tail
      double xx = 1.0, yy = 0.0;
      while (-2.0 * yy < xx * xx) {
        double a = lbft_asdouble((1023ULL << 52) | (rng.next_u64() >> 12)) - (1.0 - 0x1p-53);  // Open01
        double b = lbft_asdouble((1023ULL << 52) | (rng.next_u64() >> 12)) - (1.0 - 0x1p-53);
        xx = m.log(a) / R;
        yy = m.log(b);
      }
      return u < 0.0 ? xx - R : R - xx;
    }
    double f01 = (double)(rng.next_u64() >> 11) * 0x1p-53;  // rng.gen::<f64>() (Standard)
    if (zig_f(i + 1) + (zig_f(i) - zig_f(i + 1)) * f01 < m.exp(-x * x / 2.0)) return x;
  }
}

inline i64 f64_to_i64_sat(double v) {  // Rust `as i64`
  if (v != v) return 0;
  if (v >= 9223372036854775808.0) return INT64_MAX;
  if (v <= -9223372036854775808.0) return INT64_MIN;
  return (i64)v;
}

// simulator.rs:39-43,98-118 (+ the uniform extension)
struct RandomDelay {
  int model = 0;
  double mu = 0, sigma = 0;
  i64 lo = 0, hi = 0;
  Math math{0};
  static RandomDelay make(const lbft_oracle_config& c) {
    RandomDelay d;
    d.model = (int)c.delay_model;
    d.math = Math{(int)c.math_mode};
    // simulator.rs:101-102 (always host libm: computed once, on the host, in the product too)
    d.mu = std::log(c.mean / std::sqrt(1.0 + c.variance / (c.mean * c.mean)));
    d.sigma = std::sqrt(std::log(1.0 + c.variance / (c.mean * c.mean)));
    d.lo = c.uniform_lo;
    d.hi = c.uniform_hi;
    return d;
  }
  i64 sample(Xoshiro& rng) const {
    if (model == 1) return lo + (i64)rng.gen_range_u64((u64)(hi - lo) + 1);
    double n = standard_normal(rng, math);
    double v = math.exp(mu + sigma * n);  // LogNormal::sample = Normal(mu, sigma).sample().exp()
    return f64_to_i64_sat(v);             // `v as i64` (simulator.rs:117)
  }
};

// ------------------------------------------------------------------------------------------------
// base types (base_types.rs)
// ------------------------------------------------------------------------------------------------
constexpr i64 NEVER = INT64_MAX;  // NodeTime::never (base_types.rs:57-59)
inline i64 add_time(i64 a, i64 b) { return (i64)((u64)a + (u64)b); }  // release-mode wrapping add

using Author = u64;
using OptState = std::optional<u64>;

// ------------------------------------------------------------------------------------------------
// EpochConfiguration (configuration.rs:18-76)
// ------------------------------------------------------------------------------------------------
struct EpochConfiguration {
  std::vector<std::pair<Author, u64>> authors;
  std::unordered_map<Author, u64> voting_rights;
  u64 total_votes = 0;
  EpochConfiguration() {}
  explicit EpochConfiguration(std::vector<std::pair<Author, u64>> a) : authors(std::move(a)) {
    for (auto& p : authors) { voting_rights[p.first] = p.second; total_votes += p.second; }
  }
  u64 weight(Author a) const {  // :39-41
    auto it = voting_rights.find(a);
    return it == voting_rights.end() ? 0 : it->second;
  }
  u64 quorum_threshold() const { return 2 * total_votes / 3 + 1; }  // :52-56
  Author pick_author(u64 seed) const {                               // :65-75
    Xoshiro rng(seed);
    u64 target = rng.gen_range_u64(total_votes);
    for (auto& p : authors) {
      if (p.second > target) return p.first;
      target -= p.second;
    }
    panic("pick_author: unreachable");
  }
};

// ------------------------------------------------------------------------------------------------
// Records (record.rs:45-111) with BCS + "Name::" hashing (smr_context.rs:84-95,
// simulated_context.rs:238-242).  Signature = (author, hash) (simulated_context.rs:22-23,259-261).
// ------------------------------------------------------------------------------------------------
struct Signature { u64 author = 0, hash = 0; };

struct Block {
  u64 cmd_proposer, cmd_index;  // Command (simulated_context.rs:31-35)
  i64 time;
  u64 previous_qc_hash;
  u64 round;
  Author author;
  Signature signature;
  u64 hash() const {
    Bytes b; b.raw("Block_::");
    b.u64le(cmd_proposer); b.u64le(cmd_index); b.u64le((u64)time); b.u64le(previous_qc_hash);
    b.u64le(round); b.u64le(author);
    return b.hash();
  }
};
inline void bcs_opt(Bytes& b, const OptState& s) {
  if (s) { b.byte(1); b.u64le(*s); } else b.byte(0);
}
struct Vote {
  u64 epoch_id, round, certified_block_hash, state;
  OptState committed_state;
  Author author;
  Signature signature;
  u64 hash() const {
    Bytes b; b.raw("Vote_::");
    b.u64le(epoch_id); b.u64le(round); b.u64le(certified_block_hash); b.u64le(state);
    bcs_opt(b, committed_state); b.u64le(author);
    return b.hash();
  }
};
struct QuorumCertificate {
  u64 epoch_id, round, certified_block_hash, state;
  OptState committed_state;
  std::vector<std::pair<Author, Signature>> votes;
  Author author;
  Signature signature;
  u64 hash() const {
    Bytes b; b.raw("QuorumCertificate_::");
    b.u64le(epoch_id); b.u64le(round); b.u64le(certified_block_hash); b.u64le(state);
    bcs_opt(b, committed_state);
    b.uleb(votes.size());
    for (auto& v : votes) { b.u64le(v.first); b.u64le(v.second.author); b.u64le(v.second.hash); }
    b.u64le(author);
    return b.hash();
  }
};
struct Timeout {
  u64 epoch_id, round, highest_certified_block_round;
  Author author;
  Signature signature;
  u64 hash() const {
    Bytes b; b.raw("Timeout_::");
    b.u64le(epoch_id); b.u64le(round); b.u64le(highest_certified_block_round); b.u64le(author);
    return b.hash();
  }
};
struct Record {
  enum Kind { BLOCK, VOTE, QC, TIMEOUT } kind;
  Block block; Vote vote; QuorumCertificate qc; Timeout timeout;
  static Record of(const Block& b) { Record r; r.kind = BLOCK; r.block = b; return r; }
  static Record of(const Vote& v) { Record r; r.kind = VOTE; r.vote = v; return r; }
  static Record of(const QuorumCertificate& q) { Record r; r.kind = QC; r.qc = q; return r; }
  static Record of(const Timeout& t) { Record r; r.kind = TIMEOUT; r.timeout = t; return r; }
};
u64 epoch_id_hash(u64 id) {  // node.rs:116-118 context.hash(&EpochId)
  Bytes b; b.raw("EpochId::"); b.u64le(id);
  return b.hash();
}

// ------------------------------------------------------------------------------------------------
// SimulatedContext (simulated_context.rs:37-217)
// ------------------------------------------------------------------------------------------------
struct HistEntry { u64 proposer, index; i64 time; };
struct LedgerState {
  std::vector<HistEntry> history;
  u64 cached_key = 0;
  void rekey() {  // :51-55  Vec<(Command, NodeTime)>::hash = len word, then 3 words per entry
    Bytes b;
    b.u64le(history.size());
    for (auto& e : history) { b.u64le(e.proposer); b.u64le(e.index); b.u64le((u64)e.time); }
    cached_key = b.hash();
  }
  bool happened_just_before(const LedgerState& o) const {  // :61-71
    if (history.size() + 1 != o.history.size()) return false;
    for (size_t i = 0; i < history.size(); i++) {
      const HistEntry &a = history[i], &b = o.history[i];
      if (a.proposer != b.proposer || a.index != b.index || a.time != b.time) return false;
    }
    return true;
  }
};

struct Context {
  Author author;
  size_t num_nodes;
  u64 max_command_per_epoch;
  u64 next_fetched_command_index = 0;
  LedgerState last_committed;
  std::unordered_map<u64, LedgerState> pending;
  std::vector<u64> rights;  // extension; all 1 in the reference
  u64 rights_rotation = 0;  // extension: the rights of epoch e are rights[(i + e * rights_rotation) % n] (lbft_oracle.h)

  Context(Author a, size_t n, u64 max_cmd, std::vector<u64> r)
      : author(a), num_nodes(n), max_command_per_epoch(max_cmd), rights(std::move(r)) {
    last_committed.rekey();
  }
  const LedgerState* get_ledger_state(u64 state) const {  // :102-108
    if (state == last_committed.cached_key) return &last_committed;
    auto it = pending.find(state);
    return it == pending.end() ? nullptr : &it->second;
  }
  std::pair<u64, u64> fetch() {  // :116-125
    return {author, next_fetched_command_index++};
  }
  OptState compute(u64 base_state, u64 proposer, u64 index, i64 time) {  // :127-158
    const LedgerState* base = get_ledger_state(base_state);
    if (!base) return std::nullopt;
    LedgerState ns = *base;
    ns.history.push_back(HistEntry{proposer, index, time});
    ns.rekey();
    u64 k = ns.cached_key;
    pending[k] = std::move(ns);
    return k;
  }
  void commit(u64 state, const OptState* certificate_committed_state) {  // :160-185
    auto it = pending.find(state);
    if (it == pending.end()) panic("Committed states should be known");
    LedgerState ls = std::move(it->second);
    pending.erase(it);
    if (!last_committed.happened_just_before(ls)) panic("commit: not a successor state");
    if (certificate_committed_state && *certificate_committed_state &&
        **certificate_committed_state != state)
      panic("commit certificate mismatch");
    last_committed = std::move(ls);
  }
  u64 last_committed_state() const { return last_committed.cached_key; }  // :194-196
  u64 read_epoch_id(u64 state) const {                                    // :199-207
    const LedgerState* s = get_ledger_state(state);
    if (!s) panic("Read states should be known");
    return s->history.size() / max_command_per_epoch;
  }
  EpochConfiguration configuration(u64 state) const {  // :209-216 ("We do not simulate changes in the voting rights yet.")
    std::vector<std::pair<Author, u64>> v;
    u64 shift = rights_rotation ? (read_epoch_id(state) * rights_rotation) % num_nodes : 0;
    for (size_t i = 0; i < num_nodes; i++) v.push_back({(Author)i, rights[(i + shift) % num_nodes]});
    return EpochConfiguration(v);
  }
  // simulated_context.rs:244-261
  static bool verify(Author a, u64 hash, const Signature& s) { return a == s.author && hash == s.hash; }
  Signature sign(u64 hash) const { return Signature{author, hash}; }
};

// ------------------------------------------------------------------------------------------------
// RecordStoreState (record_store.rs:93-843)
// ------------------------------------------------------------------------------------------------
struct PacemakerView {  // what RecordStore::proposed_block needs from the pacemaker (pacemaker.rs:51-54)
  u64 active_epoch; u64 active_round; std::optional<Author> active_leader;
};

struct RecordStore {
  u64 epoch_id;
  EpochConfiguration configuration;
  u64 initial_hash;
  u64 initial_state;
  std::unordered_map<u64, Block> blocks;
  std::unordered_map<u64, QuorumCertificate> quorum_certificates;
  std::optional<u64> current_proposed_block;
  u64 highest_quorum_certificate_round = 0;
  u64 highest_quorum_certificate_hash;
  u64 highest_timeout_certificate_round = 0;
  u64 current_round = 1;
  u64 highest_committed_round = 0;
  std::optional<u64> highest_commit_certificate_hash;
  std::optional<std::vector<Timeout>> highest_timeout_certificate;
  std::map<Author, Timeout> current_timeouts;  // HashMap in the reference; ascending order here (Q4)
  std::map<Author, Vote> current_votes;
  u64 current_timeouts_weight = 0;
  enum { ONGOING, WON, CLOSED } election = ONGOING;  // :125-134
  std::map<std::pair<u64, u64>, u64> ballot;
  u64 won_block = 0, won_state = 0;
  u64 inserted_ok = 0;  // diagnostics: successful inserts

  RecordStore(u64 ih, u64 is, u64 eid, EpochConfiguration cfg)  // :169-198
      : epoch_id(eid), configuration(std::move(cfg)), initial_hash(ih), initial_state(is),
        highest_quorum_certificate_hash(ih) {}

  const Block* block(u64 h) const {  // :758-760
    auto it = blocks.find(h);
    return it == blocks.end() ? nullptr : &it->second;
  }
  const QuorumCertificate* quorum_certificate(u64 h) const {  // :419-424
    auto it = quorum_certificates.find(h);
    return it == quorum_certificates.end() ? nullptr : &it->second;
  }
  // BackwardQuorumCertificateIterator (:137-166)
  struct BackIter {
    const RecordStore* s; u64 cur;
    const QuorumCertificate* next() {
      if (cur == s->initial_hash) return nullptr;
      const QuorumCertificate* qc = s->quorum_certificate(cur);
      if (!qc) panic("backward iterator: unknown QC");
      const Block* b = s->block(qc->certified_block_hash);
      if (!b) panic("backward iterator: unknown block");
      cur = b->previous_qc_hash;
      return qc;
    }
  };
  void update_current_round(u64 round) {  // :207-219
    if (round <= current_round) return;
    current_round = round;
    current_proposed_block.reset();
    current_timeouts.clear();
    current_votes.clear();
    current_timeouts_weight = 0;
    election = ONGOING;
    ballot.clear();
  }
  void update_commit_3chain_round(u64 qc_hash) {  // :221-235
    BackIter it{this, qc_hash};
    const QuorumCertificate* q3 = it.next();
    const QuorumCertificate* q2 = it.next();
    const QuorumCertificate* q1 = it.next();
    if (q1 && q2 && q3) {
      u64 r1 = q1->round, r2 = q2->round, r3 = q3->round;
      if (r3 == r2 + 1 && r2 == r1 + 1 && r1 > highest_committed_round) {
        highest_committed_round = r1;
        highest_commit_certificate_hash = qc_hash;
      }
    }
  }
  OptState vote_committed_state(u64 block_hash) const {  // :237-255
    const Block* b = block(block_hash);
    if (!b) panic("vote_committed_state: unknown block");
    u64 r3 = b->round;
    BackIter it{this, b->previous_qc_hash};
    const QuorumCertificate* qc2 = it.next();
    const QuorumCertificate* qc1 = it.next();
    if (qc1 && qc2) {
      u64 r2 = qc2->round, r1 = qc1->round;
      if (r3 == r2 + 1 && r2 == r1 + 1) return qc1->state;
    }
    return std::nullopt;
  }
  // :257-417.  Returns false where the reference returns Err.
  bool verify_network_record(const Record& rec, u64& hash_out) const {
    switch (rec.kind) {
      case Record::BLOCK: {
        const Block& b = rec.block;
        u64 hash = b.hash();
        if (blocks.count(hash)) return false;
        if (!Context::verify(b.author, hash, b.signature)) return false;
        if (!(b.previous_qc_hash == initial_hash || quorum_certificates.count(b.previous_qc_hash)))
          return false;
        if (initial_hash == b.previous_qc_hash) {
          if (!(b.round > 0)) return false;
        } else {
          const QuorumCertificate* pqc = quorum_certificate(b.previous_qc_hash);
          const Block* pb = block(pqc->certified_block_hash);
          if (!pb) panic("verify block: previous block unknown");
          if (!(b.round > pb->round)) return false;
        }
        hash_out = hash;
        return true;
      }
      case Record::VOTE: {
        const Vote& v = rec.vote;
        u64 hash = v.hash();
        if (v.epoch_id != epoch_id) return false;
        const Block* b = block(v.certified_block_hash);
        if (!b) return false;
        if (b->round != v.round) return false;
        if (vote_committed_state(v.certified_block_hash) != v.committed_state) return false;
        if (v.round != current_round) return false;
        if (current_votes.count(v.author)) return false;
        if (!Context::verify(v.author, hash, v.signature)) return false;
        hash_out = hash;
        return true;
      }
      case Record::QC: {
        const QuorumCertificate& q = rec.qc;
        u64 hash = q.hash();
        if (q.epoch_id != epoch_id) return false;
        if (quorum_certificates.count(hash)) return false;
        const Block* b = block(q.certified_block_hash);
        if (!b) return false;
        if (b->round != q.round) return false;
        if (q.author != b->author) return false;
        if (vote_committed_state(q.certified_block_hash) != q.committed_state) return false;
        u64 weight = 0;
        for (auto& av : q.votes) {
          Vote ov;
          ov.epoch_id = epoch_id; ov.round = q.round; ov.certified_block_hash = q.certified_block_hash;
          ov.state = q.state; ov.committed_state = q.committed_state; ov.author = av.first;
          if (!Context::verify(av.first, ov.hash(), av.second)) return false;
          weight += configuration.weight(av.first);
        }
        if (!(weight >= configuration.quorum_threshold())) return false;
        if (!Context::verify(q.author, hash, q.signature)) return false;
        hash_out = hash;
        return true;
      }
      case Record::TIMEOUT: {
        const Timeout& t = rec.timeout;
        u64 hash = t.hash();
        if (t.epoch_id != epoch_id) return false;
        if (!(t.highest_certified_block_round <= highest_quorum_certificate_round)) return false;
        if (t.round != current_round) return false;
        if (current_timeouts.count(t.author)) return false;
        if (!Context::verify(t.author, hash, t.signature)) return false;
        hash_out = hash;
        return true;
      }
    }
    return false;
  }
  OptState compute_state(u64 block_hash, Context& ctx) const {  // :426-454
    const Block* b = block(block_hash);
    if (!b) panic("compute_state: unknown block");
    u64 previous_state;
    if (b->previous_qc_hash == initial_hash) previous_state = initial_state;
    else {
      const QuorumCertificate* pqc = quorum_certificate(b->previous_qc_hash);
      if (!pqc) panic("compute_state: unknown previous QC");
      previous_state = pqc->state;
    }
    return ctx.compute(previous_state, b->cmd_proposer, b->cmd_index, b->time);
  }
  static Author leader_of(const RecordStore& s, u64 round);  // pacemaker.rs:100-109

  bool try_insert_network_record(const Record& rec, Context& ctx) {  // :456-541
    u64 hash;
    if (!verify_network_record(rec, hash)) return false;
    switch (rec.kind) {
      case Record::BLOCK: {
        const Block& b = rec.block;
        if (b.round == current_round && leader_of(*this, b.round) == b.author)
          current_proposed_block = hash;
        blocks[hash] = b;
        break;
      }
      case Record::VOTE: {
        const Vote& v = rec.vote;
        current_votes[v.author] = v;
        if (election == ONGOING) {
          u64& entry = ballot[{v.certified_block_hash, v.state}];
          entry += configuration.weight(v.author);
          if (entry >= configuration.quorum_threshold()) {
            election = WON;
            won_block = v.certified_block_hash;
            won_state = v.state;
          }
        }
        break;
      }
      case Record::QC: {
        const QuorumCertificate& q = rec.qc;
        u64 block_hash = q.certified_block_hash, qc_round = q.round, qc_state = q.state;
        quorum_certificates[hash] = q;  // Q3: stored before the execution check
        OptState st = compute_state(block_hash, ctx);
        if (!st) return false;                   // bail!(...)
        if (*st != qc_state) return false;       // ensure!(state == qc_state)
        if (qc_round > highest_quorum_certificate_round) {
          highest_quorum_certificate_round = qc_round;
          highest_quorum_certificate_hash = hash;
        }
        update_current_round(qc_round + 1);
        update_commit_3chain_round(hash);
        break;
      }
      case Record::TIMEOUT: {
        const Timeout& t = rec.timeout;
        current_timeouts[t.author] = t;
        current_timeouts_weight += configuration.weight(t.author);
        if (current_timeouts_weight >= configuration.quorum_threshold()) {
          std::vector<Timeout> tc;
          for (auto& kv : current_timeouts) tc.push_back(kv.second);
          highest_timeout_certificate = std::move(tc);
          highest_timeout_certificate_round = current_round;
          update_current_round(current_round + 1);
        }
        break;
      }
    }
    return true;
  }
  void insert_network_record(const Record& rec, Context& ctx) {  // :833-842 (errors swallowed)
    if (try_insert_network_record(rec, ctx)) inserted_ok++;
  }

  // ---- RecordStore trait (:544-843) ----
  std::vector<std::pair<u64, u64>> committed_states_after(u64 after_round) const {  // :557-574
    u64 cc = highest_commit_certificate_hash ? *highest_commit_certificate_hash : initial_hash;
    BackIter it{this, cc};
    it.next(); it.next();
    std::vector<std::pair<u64, u64>> commits;
    while (const QuorumCertificate* qc = it.next()) {
      if (qc->round <= after_round) break;
      commits.push_back({qc->round, qc->state});
    }
    std::reverse(commits.begin(), commits.end());
    return commits;
  }
  u64 previous_round(u64 block_hash) const {  // :588-598
    const Block* b = block(block_hash);
    if (b->previous_qc_hash == initial_hash) return 0;
    const QuorumCertificate* qc = quorum_certificate(b->previous_qc_hash);
    return block(qc->certified_block_hash)->round;
  }
  u64 second_previous_round(u64 block_hash) const {  // :600-609
    const Block* b = block(block_hash);
    if (b->previous_qc_hash == initial_hash) return 0;
    const QuorumCertificate* qc = quorum_certificate(b->previous_qc_hash);
    return previous_round(qc->certified_block_hash);
  }
  struct Proposed { u64 hash; u64 round; Author author; };
  std::optional<Proposed> proposed_block(const PacemakerView& pm) const {  // :611-634
    if (epoch_id != pm.active_epoch || current_round != pm.active_round) return std::nullopt;
    if (!pm.active_leader) return std::nullopt;
    if (!current_proposed_block) return std::nullopt;
    const Block* b = block(*current_proposed_block);
    if (b->round != current_round) panic("proposed_block: round mismatch");
    if (b->author != *pm.active_leader) panic("proposed_block: leader mismatch");
    return Proposed{*current_proposed_block, b->round, b->author};
  }
  void create_timeout(Author author, u64 round, Context& ctx) {  // :636-649
    Timeout t;
    t.epoch_id = epoch_id; t.round = round;
    t.highest_certified_block_round = highest_quorum_certificate_round; t.author = author;
    if (author != ctx.author) panic("SignedValue::make: author mismatch");
    t.signature = ctx.sign(t.hash());
    insert_network_record(Record::of(t), ctx);
  }
  bool has_timeout(Author author, u64 round) const {  // :651-653
    return round == current_round && current_timeouts.count(author);
  }
  void propose_block(Context& ctx, u64 previous_qc_hash, i64 time) {  // :655-674
    auto cmd = ctx.fetch();
    Block b;
    b.cmd_proposer = cmd.first; b.cmd_index = cmd.second; b.time = time;
    b.previous_qc_hash = previous_qc_hash; b.round = current_round; b.author = ctx.author;
    b.signature = ctx.sign(b.hash());
    insert_network_record(Record::of(b), ctx);
  }
  bool create_vote(Context& ctx, u64 certified_block_hash) {  // :676-700
    OptState committed_state = vote_committed_state(certified_block_hash);
    OptState st = compute_state(certified_block_hash, ctx);
    if (!st) return false;
    Vote v;
    v.epoch_id = epoch_id; v.round = block(certified_block_hash)->round;
    v.certified_block_hash = certified_block_hash; v.state = *st; v.author = ctx.author;
    v.committed_state = committed_state;
    v.signature = ctx.sign(v.hash());
    insert_network_record(Record::of(v), ctx);
    return true;
  }
  bool check_for_new_quorum_certificate(Context& ctx) {  // :702-738
    if (election != WON) return false;
    if (block(won_block)->author != ctx.author) return false;
    OptState committed_state = vote_committed_state(won_block);
    QuorumCertificate q;
    for (auto& kv : current_votes)
      if (kv.second.state == won_state) q.votes.push_back({kv.second.author, kv.second.signature});
    q.epoch_id = epoch_id; q.round = current_round; q.certified_block_hash = won_block;
    q.state = won_state; q.committed_state = committed_state; q.author = ctx.author;
    q.signature = ctx.sign(q.hash());
    election = CLOSED;
    insert_network_record(Record::of(q), ctx);
    return true;
  }
  const QuorumCertificate* highest_commit_certificate() const {  // :740-743
    if (!highest_commit_certificate_hash) return nullptr;
    const QuorumCertificate* q = quorum_certificate(*highest_commit_certificate_hash);
    if (!q) panic("hcc unknown");
    return q;
  }
  const QuorumCertificate* highest_quorum_certificate() const {  // :745-747
    return quorum_certificate(highest_quorum_certificate_hash);
  }
  std::vector<Timeout> timeouts() const {  // :749-756
    std::vector<Timeout> t;
    if (highest_timeout_certificate) t = *highest_timeout_certificate;
    for (auto& kv : current_timeouts) t.push_back(kv.second);
    return t;
  }
  const Vote* current_vote(Author a) const {  // :762-764
    auto it = current_votes.find(a);
    return it == current_votes.end() ? nullptr : &it->second;
  }
  static bool is_power2_minus1(size_t x) { return (x & (x + 1)) == 0; }  // util.rs:8-10
  std::set<u64> known_quorum_certificate_rounds() const {                // :766-799
    std::set<u64> result;
    u64 starts[2] = {highest_quorum_certificate_hash,
                     highest_commit_certificate_hash ? *highest_commit_certificate_hash : initial_hash};
    for (u64 start : starts) {
      BackIter it{this, start};
      size_t i = 0;
      while (const QuorumCertificate* qc = it.next()) {
        if (is_power2_minus1(i)) result.insert(qc->round);
        i++;
      }
    }
    return result;
  }
  std::vector<Record> unknown_records(const std::set<u64>& known) const {  // :801-831
    auto chain = [&](u64 start) {
      std::vector<const QuorumCertificate*> c;
      BackIter it{this, start};
      while (const QuorumCertificate* qc = it.next()) {
        if (known.count(qc->round)) break;
        c.push_back(qc);
      }
      return c;
    };
    auto c1 = chain(highest_quorum_certificate_hash);
    auto c2 = chain(highest_commit_certificate_hash ? *highest_commit_certificate_hash : initial_hash);
    // util.rs:12-53 merge_sort with cmp = qc2.round.cmp(qc1.round) (descending rounds)
    std::vector<const QuorumCertificate*> qcs;
    size_t a = 0, b = 0;
    while (a < c1.size() && b < c2.size()) {
      u64 r1 = c1[a]->round, r2 = c2[b]->round;
      if (r2 < r1) qcs.push_back(c1[a++]);          // Ordering::Less
      else if (r2 == r1) {                           // Equal
        if (c1[a] == c2[b]) qcs.push_back(c1[a]);    // same reference => same QC
        else { qcs.push_back(c1[a]); qcs.push_back(c2[b]); }
        a++; b++;
      } else qcs.push_back(c2[b++]);                 // Greater
    }
    while (a < c1.size()) qcs.push_back(c1[a++]);
    while (b < c2.size()) qcs.push_back(c2[b++]);
    std::vector<Record> result;
    for (size_t n = qcs.size(); n-- > 0;) {
      result.push_back(Record::of(*block(qcs[n]->certified_block_hash)));
      result.push_back(Record::of(*qcs[n]));
    }
    for (auto& t : timeouts()) result.push_back(Record::of(t));
    if (current_proposed_block) result.push_back(Record::of(*block(*current_proposed_block)));
    return result;
  }
};

Author RecordStore::leader_of(const RecordStore& s, u64 round) {  // pacemaker.rs:100-109
  Bytes b; b.u64le(round);                                        // Round(usize)::hash
  return s.configuration.pick_author(b.hash());
}

// ------------------------------------------------------------------------------------------------
// Pacemaker (pacemaker.rs)
// ------------------------------------------------------------------------------------------------
struct PacemakerActions {  // :18-31,127-138
  std::optional<u64> should_propose_block;
  std::optional<u64> should_create_timeout;
  std::vector<Author> should_send;
  bool should_broadcast = false;
  bool should_query_all = false;
  i64 next_scheduled_update = NEVER;
};
struct PacemakerState {  // :60-77
  u64 active_epoch; u64 active_round = 0; std::optional<Author> active_leader;
  i64 active_round_start_time; i64 active_round_duration = 0;
  i64 delta; double gamma; double lambda;
  PacemakerView view() const { return PacemakerView{active_epoch, active_round, active_leader}; }
  i64 duration(const RecordStore& rs, u64 round) const {  // :111-124
    u64 hccr = rs.highest_committed_round > 0 ? rs.highest_committed_round + 2 : 0;
    if (!(round > hccr)) panic("Active round is higher than any QC round.");
    u64 n = round - hccr;
    return f64_to_i64_sat((double)delta * std::pow((double)n, gamma));
  }
  PacemakerActions update_pacemaker(Author local_author, u64 epoch_id, const RecordStore& rs,
                                    i64 latest_query_all_time, i64 clock) {  // :142-207
    PacemakerActions actions;
    u64 ar = std::max(rs.highest_quorum_certificate_round, rs.highest_timeout_certificate_round) + 1;
    if (epoch_id > active_epoch || (epoch_id == active_epoch && ar > active_round)) {
      active_epoch = epoch_id;
      active_round = ar;
      active_round_start_time = clock;
      active_leader = RecordStore::leader_of(rs, ar);
      active_round_duration = duration(rs, ar);
      if (active_leader != std::optional<Author>(local_author)) actions.should_send = {*active_leader};
    }
    if (active_leader == std::optional<Author>(local_author) && !rs.proposed_block(view())) {
      actions.should_propose_block = rs.highest_quorum_certificate_hash;
      actions.should_broadcast = true;
      actions.next_scheduled_update = clock;
    }
    if (!rs.has_timeout(local_author, ar)) {
      i64 timeout_deadline = add_time(active_round_start_time, active_round_duration);
      if (clock >= timeout_deadline) {
        actions.should_create_timeout = ar;
        actions.should_broadcast = true;
      } else {
        actions.next_scheduled_update = std::min(actions.next_scheduled_update, timeout_deadline);
      }
    } else {
      i64 period = f64_to_i64_sat(lambda * (double)active_round_duration);
      i64 query_all_deadline = add_time(latest_query_all_time, period);
      if (clock >= query_all_deadline) {
        actions.should_query_all = true;
        query_all_deadline = add_time(clock, period);
      }
      actions.next_scheduled_update = std::min(actions.next_scheduled_update, query_all_deadline);
    }
    return actions;
  }
};

// ------------------------------------------------------------------------------------------------
// NodeState, CommitTracker (node.rs) and the data-sync handlers (data_sync.rs)
// ------------------------------------------------------------------------------------------------
struct NodeUpdateActions {  // interfaces.rs:12-21
  i64 next_scheduled_update = NEVER;
  std::vector<Author> should_send;
  bool should_broadcast = false;
  bool should_query_all = false;
};
struct Notification {  // data_sync.rs:16-39
  u64 current_epoch;
  std::optional<QuorumCertificate> highest_commit_certificate;
  std::optional<QuorumCertificate> highest_quorum_certificate;
  std::vector<Timeout> timeouts;
  std::optional<Vote> current_vote;
  std::optional<Block> proposed_block;
};
struct Request {  // :41-47
  u64 current_epoch;
  std::set<u64> known_quorum_certificates;
};
struct Response {  // :49-59
  u64 current_epoch;
  std::vector<std::pair<u64, std::vector<Record>>> records;
};

struct CommitTracker {  // node.rs:50-71,363-397
  u64 epoch_id; u64 highest_committed_round = 0; i64 latest_commit_time; i64 target_commit_interval;
  struct Actions { i64 next_scheduled_update = NEVER; bool should_query_all = false; };
  Actions update_tracker(i64 latest_query_all_time, i64 clock, u64 current_epoch_id, const RecordStore& rs) {
    Actions actions;
    if (current_epoch_id > epoch_id) {
      epoch_id = current_epoch_id;
      highest_committed_round = rs.highest_committed_round;
      latest_commit_time = clock;
    } else {
      u64 hcr = rs.highest_committed_round;
      if (hcr > highest_committed_round) {
        highest_committed_round = hcr;
        latest_commit_time = clock;
      }
    }
    i64 deadline = add_time(std::max(latest_commit_time, latest_query_all_time), target_commit_interval);
    if (clock >= deadline) {
      actions.should_query_all = true;
      deadline = add_time(clock, target_commit_interval);
    }
    actions.next_scheduled_update = deadline;
    return actions;
  }
};

struct NodeState {  // node.rs:28-45
  std::unique_ptr<RecordStore> record_store;
  PacemakerState pacemaker;
  u64 epoch_id;
  u64 latest_voted_round = 0;
  u64 locked_round = 0;
  i64 latest_query_all_time;
  CommitTracker tracker;
  std::map<u64, std::unique_ptr<RecordStore>> past_record_stores;
  u32 quirks = 0;
  bool equivocator = false;                 // extension, see lbft_oracle.h "Equivocators"
  std::map<u64, Block> equivocation_twin;   // hash of proposed block B -> its twin A
  u64 response_inserts = 0;

  static NodeState make_initial_state(const Context& ctx, const lbft_oracle_config& c, i64 node_time) {  // :87-114
    NodeState n;
    u64 initial_state = ctx.last_committed_state();
    u64 epoch_id = ctx.read_epoch_id(initial_state);
    n.tracker = CommitTracker{epoch_id, 0, node_time, c.target_commit_interval};
    n.record_store.reset(new RecordStore(epoch_id_hash(epoch_id), initial_state, epoch_id,
                                         ctx.configuration(initial_state)));
    n.pacemaker = PacemakerState{epoch_id, 0, std::nullopt, node_time, 0, c.delta, c.gamma, c.lambda};
    n.epoch_id = epoch_id;
    n.latest_query_all_time = node_time;
    n.quirks = c.quirks;
    return n;
  }
  const RecordStore* record_store_at(u64 e) const {  // :128-135
    if (e == epoch_id) return record_store.get();
    auto it = past_record_stores.find(e);
    return it == past_record_stores.end() ? nullptr : it->second.get();
  }
  void insert_network_record(u64 e, const Record& r, Context& ctx) {  // :151-167
    if (e == epoch_id) record_store->insert_network_record(r, ctx);
  }
  void update_tracker(i64 clock) {  // :141-149
    tracker.update_tracker(latest_query_all_time, clock, epoch_id, *record_store);
  }
  NodeUpdateActions process_pacemaker_actions(const PacemakerActions& pa, i64 clock, Context& ctx) {  // :179-202
    NodeUpdateActions actions;
    actions.next_scheduled_update = pa.next_scheduled_update;
    actions.should_broadcast = pa.should_broadcast;
    actions.should_query_all = pa.should_query_all;
    actions.should_send = pa.should_send;
    if (pa.should_create_timeout) {
      record_store->create_timeout(ctx.author, *pa.should_create_timeout, ctx);
      latest_voted_round = std::max(latest_voted_round, *pa.should_create_timeout);
    }
    if (pa.should_propose_block) {
      if (equivocator) {  // (E1): A first, then B on the same previous QC
        record_store->propose_block(ctx, *pa.should_propose_block, clock);
        std::optional<u64> a = record_store->current_proposed_block;
        record_store->propose_block(ctx, *pa.should_propose_block, clock);
        std::optional<u64> b = record_store->current_proposed_block;
        if (a && b && *a != *b) equivocation_twin[*b] = *record_store->block(*a);
      } else {
        record_store->propose_block(ctx, *pa.should_propose_block, clock);
      }
    }
    return actions;
  }
  void process_commits(Context& ctx) {  // :313-350
    for (auto& rs : record_store->committed_states_after(tracker.highest_committed_round)) {
      u64 round = rs.first, state = rs.second;
      if (round == record_store->highest_committed_round) {
        const QuorumCertificate* hcc = record_store->highest_commit_certificate();
        if (!hcc) ctx.commit(state, nullptr);
        else ctx.commit(state, &hcc->committed_state);
      } else ctx.commit(state, nullptr);
      u64 new_epoch_id = ctx.read_epoch_id(state);
      if (new_epoch_id > epoch_id) {
        std::unique_ptr<RecordStore> nrs(new RecordStore(epoch_id_hash(new_epoch_id), state, new_epoch_id,
                                                          ctx.configuration(state)));
        past_record_stores[epoch_id] = std::move(record_store);
        record_store = std::move(nrs);
        epoch_id = new_epoch_id;
        latest_voted_round = 0;
        locked_round = 0;
        break;
      }
    }
  }
  NodeUpdateActions update_node(Context& ctx, i64 clock) {  // :240-304
    PacemakerActions pa = pacemaker.update_pacemaker(ctx.author, epoch_id, *record_store,
                                                     latest_query_all_time, clock);
    NodeUpdateActions actions = process_pacemaker_actions(pa, clock, ctx);
    if (auto pb = record_store->proposed_block(pacemaker.view())) {
      if (pb->round > latest_voted_round && record_store->previous_round(pb->hash) >= locked_round) {
        latest_voted_round = pb->round;
        locked_round = std::max(locked_round, record_store->second_previous_round(pb->hash));
        if (record_store->create_vote(ctx, pb->hash)) actions.should_send = {pb->author};
      }
    }
    if (record_store->check_for_new_quorum_certificate(ctx)) {
      actions.should_broadcast = true;
      actions.next_scheduled_update = clock;
    }
    process_commits(ctx);
    CommitTracker::Actions ta = tracker.update_tracker(latest_query_all_time, clock, epoch_id, *record_store);
    actions.should_query_all = actions.should_query_all || ta.should_query_all;
    actions.next_scheduled_update = std::min(actions.next_scheduled_update, ta.next_scheduled_update);
    if (actions.should_query_all) latest_query_all_time = clock;
    return actions;
  }

  // ---- data_sync.rs ----
  Request create_request_internal() const {  // :66-71
    return Request{epoch_id, record_store->known_quorum_certificate_rounds()};
  }
  Notification create_notification(const Context& ctx) const {  // :82-111
    Notification n;
    const QuorumCertificate* hcc = record_store->highest_commit_certificate();
    if (hcc) n.highest_commit_certificate = *hcc;
    else {
      // EpochId::previous (base_types.rs:31-37) returns Some(self) for id > 0 (quirk Q2)
      std::optional<u64> prev;
      if (epoch_id != 0) prev = (quirks & 2) ? epoch_id - 1 : epoch_id;
      if (prev) {
        const RecordStore* ps = record_store_at(*prev);
        if (!ps) panic("The record store of the previous epoch should exist.");
        const QuorumCertificate* p = ps->highest_commit_certificate();
        if (p) n.highest_commit_certificate = *p;
      }
    }
    n.current_epoch = epoch_id;
    if (const QuorumCertificate* q = record_store->highest_quorum_certificate()) n.highest_quorum_certificate = *q;
    n.timeouts = record_store->timeouts();
    if (const Vote* v = record_store->current_vote(ctx.author)) n.current_vote = *v;
    if (auto pb = record_store->proposed_block(pacemaker.view())) {
      if (pb->author == ctx.author) n.proposed_block = *record_store->block(pb->hash);
    }
    return n;
  }
  // (E2): the copy of `n` that even-indexed receivers get, if `n` carries an equivocal proposal
  std::optional<Notification> equivocal_copy(const Notification& n) const {
    if (!equivocator || !n.proposed_block) return std::nullopt;
    auto it = equivocation_twin.find(n.proposed_block->hash());
    if (it == equivocation_twin.end()) return std::nullopt;
    Notification alt = n;
    alt.proposed_block = it->second;
    return alt;
  }
  std::optional<Request> handle_notification(Context& ctx, const Notification& n) {  // :113-177
    bool should_sync = false;
    should_sync |= n.current_epoch > epoch_id;
    if (n.highest_commit_certificate) {
      const QuorumCertificate& q = *n.highest_commit_certificate;
      insert_network_record(q.epoch_id, Record::of(q), ctx);
      should_sync |= (q.epoch_id > epoch_id) ||
                     (q.epoch_id == epoch_id && q.round > record_store->highest_committed_round + 2);
    }
    if (n.highest_quorum_certificate) {
      const QuorumCertificate& q = *n.highest_quorum_certificate;
      insert_network_record(q.epoch_id, Record::of(q), ctx);
      should_sync |= (q.epoch_id > epoch_id) ||
                     (q.epoch_id == epoch_id && q.round > record_store->highest_quorum_certificate_round);
    }
    if (n.proposed_block) insert_network_record(n.current_epoch, Record::of(*n.proposed_block), ctx);
    for (auto& t : n.timeouts) insert_network_record(n.current_epoch, Record::of(t), ctx);
    if (n.current_vote) insert_network_record(n.current_epoch, Record::of(*n.current_vote), ctx);
    if (should_sync) return create_request_internal();
    return std::nullopt;
  }
  Response handle_request(const Request& req) const {  // :183-207
    Response r;
    if (const RecordStore* s = record_store_at(req.current_epoch))
      r.records.push_back({req.current_epoch, s->unknown_records(req.known_quorum_certificates)});
    for (u64 i = req.current_epoch + 1; i < epoch_id + 1; i++) {
      const RecordStore* s = record_store_at(i);
      if (!s) panic("All record stores up to the current epoch should exist.");
      r.records.push_back({i, s->unknown_records(std::set<u64>())});
    }
    r.current_epoch = epoch_id;
    return r;
  }
  void handle_response(Context& ctx, const Response& resp, i64 clock) {  // :209-240
    size_t num_records = resp.records.size();
    for (size_t i = 0; i < num_records; i++) {
      u64 e = resp.records[i].first;
      if (e < epoch_id) continue;
      if (e > epoch_id) break;
      u64 before = record_store->inserted_ok;
      RecordStore* rs_before = record_store.get();
      for (auto& rec : resp.records[i].second) insert_network_record(e, rec, ctx);
      if (record_store.get() == rs_before) response_inserts += record_store->inserted_ok - before;
      if (i == num_records - 1) break;
      process_commits(ctx);
      update_tracker(clock);
    }
  }
};

// ------------------------------------------------------------------------------------------------
// Simulator (simulator.rs)
// ------------------------------------------------------------------------------------------------
struct Event {  // :47-82
  i64 scheduled_time;
  u64 creation_stamp;
  int kind;  // 0 notify, 1 request, 2 response, 3 timer (:129-139)
  Author receiver = 0, sender = 0, author = 0;
  std::shared_ptr<Notification> notification;
  std::shared_ptr<Request> request;
  std::shared_ptr<Response> response;
};
struct EventLess {  // Ord for ScheduledEvent (:149-161); std::priority_queue is a max-heap like BinaryHeap
  bool operator()(const Event& a, const Event& b) const {
    // a < b  <=>  (b.time, a.kind, b.stamp) < (a.time, b.kind, a.stamp)
    if (b.scheduled_time != a.scheduled_time) return b.scheduled_time < a.scheduled_time;
    if (a.kind != b.kind) return a.kind < b.kind;
    return b.creation_stamp < a.creation_stamp;
  }
};
struct SimNode {  // :53-59
  i64 startup_time;
  i64 ignore_scheduled_updates_until;
  NodeState node;
  Context context;
};

}  // namespace

static size_t node_state_image_size(const NodeState& n);  // bincode::serialize(&NodeState) (defined with lbft_oracle_save_node)

struct lbft_oracle_sim {
  lbft_oracle_config cfg;
  std::vector<u64> rights;
  i64 clock = 0;
  RandomDelay network_delay;
  std::priority_queue<Event, std::vector<Event>, EventLess> pending_events;
  std::vector<SimNode> nodes;
  u64 event_count = 0;
  Xoshiro rng;
  lbft_oracle_counters counters{};
  std::string error;
  std::vector<std::shared_ptr<Notification>> manual_notifications;  // node-level interface
  std::vector<Request> manual_requests;
  std::vector<Response> manual_responses;
  // DataWriter (bft-lib/src/data_writer.rs:10-60): first pop time at which each node was seen in each round
  bool data_writer = false;
  std::vector<u64> dw_max_round;
  std::vector<std::vector<std::pair<u64, i64>>> dw_switches;
  u64 dw_messages = 0;

  lbft_oracle_sim(const lbft_oracle_config& c, u64 seed) : cfg(c), rng(seed) {  // Simulator::new :200-250
    rights.assign(c.num_nodes, 1);
    if (c.voting_rights) rights.assign(c.voting_rights, c.voting_rights + c.num_nodes);
    cfg.voting_rights = nullptr;
    network_delay = RandomDelay::make(c);
    nodes.reserve(c.num_nodes);
    for (u32 index = 0; index < c.num_nodes; index++) {
      // context_factory (main.rs:23-34): SimulatedContext::new + NodeState::make_initial_state(.., NodeTime(0))
      Context context((Author)index, c.num_nodes, c.commands_per_epoch, rights);
      context.rights_rotation = c.rights_rotation;
      NodeState node = NodeState::make_initial_state(context, c, 0);
      node.equivocator = c.equivocate_every && index % c.equivocate_every == 0;
      i64 startup_time = clock + network_delay.sample(rng) + 1;
      i64 scheduled_time = 0 + startup_time;  // from_node_time(NodeTime(0), startup)
      Event ev{scheduled_time, event_count++, 3};
      ev.author = index;
      push(ev);
      nodes.push_back(SimNode{startup_time, startup_time - 1, std::move(node), std::move(context)});
    }
  }
  void push(const Event& e) {
    pending_events.push(e);
    counters.max_queue = std::max<u64>(counters.max_queue, pending_events.size());
  }
  void schedule_event(i64 t, Event e) {  // :252-264
    e.scheduled_time = t;
    e.creation_stamp = event_count++;
    push(e);
  }
  void schedule_network_event(Event e) {  // :266-269
    i64 t = clock + network_delay.sample(rng);
    // extension "Lossy network" (lbft_oracle.h): (L1) random loss, (L2) partition
    bool lost = false;
    if (cfg.drop_per_million) {
      u64 d = rng.next_u64();
      lost = (u64)(((unsigned __int128)d * 1000000u) >> 64) < cfg.drop_per_million;
    }
    if (cfg.partition_size && clock >= cfg.partition_start && clock < cfg.partition_end &&
        ((e.sender < cfg.partition_size) != (e.receiver < cfg.partition_size)))
      lost = true;
    if (lost) { event_count++; return; }
    schedule_event(t, std::move(e));
  }
  NodeUpdateActions node_update(SimNode& n, i64 global_clock) {  // :176-179
    return n.node.update_node(n.context, global_clock - n.startup_time);
  }
  void process_node_actions(i64 clk, Author author, const NodeUpdateActions& actions) {  // :296-378
    SimNode& node = nodes[author];
    // save_node: a semantic no-op inside a run (node.rs:233-238), but the reference pays for it on every processed event
    // (simulator.rs:307-309: bincode of the whole NodeState); reference_overheads = 1 pays it too, for an honest CPU baseline
    if (cfg.reference_overheads) counters.saved_bytes += node_state_image_size(node.node);
    i64 new_scheduled_time = std::max(add_time(actions.next_scheduled_update, node.startup_time), clk + 1);
    node.ignore_scheduled_updates_until = new_scheduled_time - 1;
    {
      Event ev{0, 0, 3};
      ev.author = author;
      schedule_event(new_scheduled_time, ev);
    }
    std::vector<Author> receivers;
    if (actions.should_broadcast) {
      for (u32 i = 0; i < nodes.size(); i++) if (i != author) receivers.push_back(i);
    } else {
      for (Author r : actions.should_send) if (r != author) receivers.push_back(r);
    }
    rng.shuffle(receivers);
    auto notification = std::make_shared<Notification>(node.node.create_notification(node.context));
    std::shared_ptr<Notification> alt;
    if (auto a = node.node.equivocal_copy(*notification)) alt = std::make_shared<Notification>(std::move(*a));
    for (Author r : receivers) {
      Event ev{0, 0, 0};
      ev.sender = author; ev.receiver = r; ev.notification = (alt && r % 2 == 0) ? alt : notification;
      if (cfg.reference_overheads) ev.notification = std::make_shared<Notification>(*ev.notification);  // notification.clone() per receiver (simulator.rs:348-354)
      schedule_network_event(ev);
    }
    std::vector<Author> senders;
    if (actions.should_query_all)
      for (u32 i = 0; i < nodes.size(); i++) if (i != author) senders.push_back(i);
    auto request = std::make_shared<Request>(node.node.create_request_internal());
    rng.shuffle(senders);
    for (Author s : senders) {
      Event ev{0, 0, 1};
      ev.receiver = author; ev.sender = s; ev.request = request;
      schedule_network_event(ev);
    }
  }
  void loop_until(i64 max_clock) {  // :380-475
    while (!pending_events.empty()) {
      Event ev = pending_events.top();
      pending_events.pop();
      if (ev.scheduled_time > max_clock) break;
      if (data_writer) {  // simulator.rs:393-396 -> data_writer.rs:34-60 (uses the event's own scheduled time)
        for (size_t k = 0; k < nodes.size(); k++) {
          u64 r = nodes[k].node.pacemaker.active_round;
          if (r > dw_max_round[k]) { dw_max_round[k] = r; dw_switches[k].push_back({r, ev.scheduled_time}); }
        }
        if (ev.kind != 3) dw_messages++;
      }
      i64 clk = std::max(ev.scheduled_time, clock);
      clock = clk;
      counters.events[ev.kind]++;
      switch (ev.kind) {
        case 3: {
          SimNode& node = nodes[ev.author];
          if (clk <= node.ignore_scheduled_updates_until) continue;
          NodeUpdateActions actions = node_update(node, clk);
          process_node_actions(clk, ev.author, actions);
          break;
        }
        case 0: {
          SimNode& node = nodes[ev.receiver];
          std::optional<Request> result = node.node.handle_notification(node.context, *ev.notification);
          NodeUpdateActions actions = node_update(node, clk);
          if (result) {
            Event e2{0, 0, 1};
            e2.sender = ev.sender; e2.receiver = ev.receiver;
            e2.request = std::make_shared<Request>(std::move(*result));
            schedule_network_event(e2);
          }
          process_node_actions(clk, ev.receiver, actions);
          break;
        }
        case 1: {
          // Q1: the reference answers the request on `receiver` itself (simulator.rs:446);
          // quirks bit0 routes it to the peer like bft-driver/src/core.rs:174-178.
          SimNode& node = nodes[(cfg.quirks & 1) ? ev.sender : ev.receiver];
          Event e2{0, 0, 2};
          e2.sender = ev.sender; e2.receiver = ev.receiver;
          e2.response = std::make_shared<Response>(node.node.handle_request(*ev.request));
          schedule_network_event(e2);
          break;
        }
        case 2: {
          SimNode& node = nodes[ev.receiver];
          i64 local_clock = clk - node.startup_time;
          node.node.handle_response(node.context, *ev.response, local_clock);
          NodeUpdateActions actions = node_update(node, clk);
          process_node_actions(clk, ev.receiver, actions);
          break;
        }
      }
    }
    finalize_counters();
  }
  void finalize_counters() {
    counters.rng_draws = rng.draws;
    counters.events_scheduled = event_count;
    u64 mr = UINT64_MAX, mc = UINT64_MAX, ri = 0;
    for (auto& n : nodes) {
      mr = std::min<u64>(mr, n.node.pacemaker.active_round);
      mc = std::min<u64>(mc, n.context.last_committed.history.size());
      ri += n.node.response_inserts;
    }
    counters.rounds = nodes.empty() ? 0 : mr;
    counters.commits = nodes.empty() ? 0 : mc;
    counters.response_inserts = ri;
  }
};

// ------------------------------------------------------------------------------------------------
// C API
// ------------------------------------------------------------------------------------------------
extern "C" {

int lbft_oracle_create(const lbft_oracle_config* cfg, uint64_t seed, lbft_oracle_sim** out) {
  if (!cfg || !out || cfg->num_nodes == 0) return -1;
  try {
    *out = new lbft_oracle_sim(*cfg, seed);
  } catch (Panic& p) {
    return -2;
  }
  return 0;
}
int lbft_oracle_run_until(lbft_oracle_sim* sim, int64_t max_clock) {
  try {
    sim->loop_until(max_clock);
  } catch (Panic& p) {
    sim->error = p.msg;
    sim->finalize_counters();
    return -3;
  }
  return 0;
}
void lbft_oracle_destroy(lbft_oracle_sim* sim) { delete sim; }
size_t lbft_oracle_commit_count(const lbft_oracle_sim* sim, uint32_t node) {
  return sim->nodes[node].context.last_committed.history.size();
}
size_t lbft_oracle_committed_history(const lbft_oracle_sim* sim, uint32_t node, lbft_oracle_commit* out, size_t cap) {
  auto& h = sim->nodes[node].context.last_committed.history;
  for (size_t i = 0; i < h.size() && i < cap; i++) out[i] = lbft_oracle_commit{h[i].proposer, h[i].index, h[i].time};
  return h.size();
}
uint64_t lbft_oracle_last_committed_state(const lbft_oracle_sim* sim, uint32_t node) {
  return sim->nodes[node].context.last_committed_state();
}
size_t lbft_oracle_committed_record_hashes(const lbft_oracle_sim* sim, uint32_t node, lbft_oracle_record_hash* out, size_t cap) {
  const SimNode& n = sim->nodes[node];
  auto& h = n.context.last_committed.history;
  // the node's record stores, oldest epoch first (a Command (proposer, index) identifies its block: one fetch per proposal)
  std::vector<const RecordStore*> stores;
  for (auto& kv : n.node.past_record_stores) stores.push_back(kv.second.get());
  stores.push_back(n.node.record_store.get());
  for (size_t k = 0; k < h.size() && k < cap; k++) {
    lbft_oracle_record_hash r{0, 0, 0, 0, 0};
    for (const RecordStore* s : stores) {
      for (auto& kv : s->blocks) {
        const Block& b = kv.second;
        if (b.cmd_proposer != h[k].proposer || b.cmd_index != h[k].index) continue;
        r.block_hash = kv.first;
        for (auto& q : s->quorum_certificates)
          if (q.second.certified_block_hash == kv.first) {
            r.qc_hash = q.first; r.state = q.second.state; r.has_qc = 1; r.num_votes = (uint32_t)q.second.votes.size();
          }
      }
    }
    out[k] = r;
  }
  return h.size();
}
// ---- ConsensusNode::save_node (node.rs:233-238): bincode::serialize(&NodeState) ---------------------------------------
// bincode 1.3 with its default options (the reference calls bincode::serialize / deserialize directly): integers fixed-width
// little-endian (usize as u64), f64 as its 8 IEEE bytes, Option = one tag byte (+ value), Vec / HashMap = u64 length then the
// elements, enum = u32 variant index then the variant's fields, structs / tuples / newtypes = their fields in order.  The
// reference's HashMaps serialise in per-process iteration order, which bincode's deserialiser does not care about;
// this image uses the CANONICAL order -- ascending key -- so that it is reproducible: load_node accepts it like any other.
namespace {
struct Bin {
  std::vector<u8> b;
  void u64v(u64 v) { for (int i = 0; i < 8; i++) b.push_back((u8)(v >> (8 * i))); }
  void i64v(i64 v) { u64v((u64)v); }
  void u32v(u32 v) { for (int i = 0; i < 4; i++) b.push_back((u8)(v >> (8 * i))); }
  void f64v(double d) { u64 u; memcpy(&u, &d, 8); u64v(u); }
  void opt(const std::optional<u64>& o) { if (o) { b.push_back(1); u64v(*o); } else b.push_back(0); }
  void sig(const Signature& g) { u64v(g.author); u64v(g.hash); }
  void block(const Block& x) { u64v(x.cmd_proposer); u64v(x.cmd_index); i64v(x.time); u64v(x.previous_qc_hash); u64v(x.round); u64v(x.author); sig(x.signature); }
  void vote(const Vote& v) { u64v(v.epoch_id); u64v(v.round); u64v(v.certified_block_hash); u64v(v.state); opt(v.committed_state); u64v(v.author); sig(v.signature); }
  void qc(const QuorumCertificate& q) {
    u64v(q.epoch_id); u64v(q.round); u64v(q.certified_block_hash); u64v(q.state); opt(q.committed_state);
    u64v(q.votes.size());
    for (auto& v : q.votes) { u64v(v.first); sig(v.second); }
    u64v(q.author); sig(q.signature);
  }
  void timeout(const Timeout& t) { u64v(t.epoch_id); u64v(t.round); u64v(t.highest_certified_block_round); u64v(t.author); sig(t.signature); }
  void store(const RecordStore& r) {  // RecordStoreState (record_store.rs:93-119), field by field
    u64v(r.epoch_id);
    u64v(r.configuration.authors.size());
    for (auto& a : r.configuration.authors) { u64v(a.first); u64v(a.second); }
    std::map<Author, u64> rights(r.configuration.voting_rights.begin(), r.configuration.voting_rights.end());
    u64v(rights.size());
    for (auto& a : rights) { u64v(a.first); u64v(a.second); }
    u64v(r.configuration.total_votes);
    u64v(r.initial_hash);
    u64v(r.initial_state);
    std::map<u64, const Block*> blocks;
    for (auto& kv : r.blocks) blocks[kv.first] = &kv.second;
    u64v(blocks.size());
    for (auto& kv : blocks) { u64v(kv.first); block(*kv.second); }
    std::map<u64, const QuorumCertificate*> qcs;
    for (auto& kv : r.quorum_certificates) qcs[kv.first] = &kv.second;
    u64v(qcs.size());
    for (auto& kv : qcs) { u64v(kv.first); qc(*kv.second); }
    opt(r.current_proposed_block);
    u64v(r.highest_quorum_certificate_round);
    u64v(r.highest_quorum_certificate_hash);
    u64v(r.highest_timeout_certificate_round);
    u64v(r.current_round);
    u64v(r.highest_committed_round);
    opt(r.highest_commit_certificate_hash);
    if (r.highest_timeout_certificate) {
      b.push_back(1);
      u64v(r.highest_timeout_certificate->size());
      for (auto& t : *r.highest_timeout_certificate) timeout(t);
    } else b.push_back(0);
    u64v(r.current_timeouts.size());
    for (auto& kv : r.current_timeouts) { u64v(kv.first); timeout(kv.second); }
    u64v(r.current_votes.size());
    for (auto& kv : r.current_votes) { u64v(kv.first); vote(kv.second); }
    u64v(r.current_timeouts_weight);
    if (r.election == RecordStore::ONGOING) {
      u32v(0);
      u64v(r.ballot.size());
      for (auto& kv : r.ballot) { u64v(kv.first.first); u64v(kv.first.second); u64v(kv.second); }
    } else if (r.election == RecordStore::WON) {
      u32v(1); u64v(r.won_block); u64v(r.won_state);
    } else u32v(2);
  }
};
}  // namespace
extern "C++" {
static void node_state_image(const NodeState& n, Bin& w) {
  w.store(*n.record_store);
  const PacemakerState& pm = n.pacemaker;  // PacemakerState (pacemaker.rs:60-77)
  w.u64v(pm.active_epoch); w.u64v(pm.active_round);
  if (pm.active_leader) { w.b.push_back(1); w.u64v(*pm.active_leader); } else w.b.push_back(0);
  w.i64v(pm.active_round_start_time); w.i64v(pm.active_round_duration); w.i64v(pm.delta); w.f64v(pm.gamma); w.f64v(pm.lambda);
  w.u64v(n.epoch_id); w.u64v(n.latest_voted_round); w.u64v(n.locked_round); w.i64v(n.latest_query_all_time);
  w.u64v(n.tracker.epoch_id); w.u64v(n.tracker.highest_committed_round); w.i64v(n.tracker.latest_commit_time); w.i64v(n.tracker.target_commit_interval);
  w.u64v(n.past_record_stores.size());
  for (auto& kv : n.past_record_stores) { w.u64v(kv.first); w.store(*kv.second); }
}
static size_t node_state_image_size(const NodeState& n) { Bin w; node_state_image(n, w); return w.b.size(); }
}  // extern "C++"
size_t lbft_oracle_save_node(const lbft_oracle_sim* sim, uint32_t node, uint8_t* out, size_t cap) {
  if (!sim || node >= sim->nodes.size()) return 0;
  Bin w;
  node_state_image(sim->nodes[node].node, w);
  if (out && cap >= w.b.size()) memcpy(out, w.b.data(), w.b.size());
  return w.b.size();
}
int lbft_oracle_node_update(lbft_oracle_sim* sim, uint32_t node, int64_t node_time, lbft_oracle_actions* out) {
  if (!sim || !out || node >= sim->nodes.size()) return -1;
  try {
    SimNode& n = sim->nodes[node];
    NodeUpdateActions a = n.node.update_node(n.context, node_time);
    out->next_scheduled_update = a.next_scheduled_update;
    out->should_send[0] = out->should_send[1] = 0;
    for (Author r : a.should_send) out->should_send[r >> 6] |= 1ULL << (r & 63);
    out->should_broadcast = a.should_broadcast;
    out->should_query_all = a.should_query_all;
    return 0;
  } catch (const Panic& p) { sim->error = p.msg; return -2; }
}
int lbft_oracle_node_create_notification(lbft_oracle_sim* sim, uint32_t node) {
  if (!sim || node >= sim->nodes.size()) return -1;
  try {
    SimNode& n = sim->nodes[node];
    sim->manual_notifications.push_back(std::make_shared<Notification>(n.node.create_notification(n.context)));
    return (int)sim->manual_notifications.size() - 1;
  } catch (const Panic& p) { sim->error = p.msg; return -2; }
}
int lbft_oracle_node_handle_notification(lbft_oracle_sim* sim, uint32_t receiver, int handle, uint32_t* should_sync) {
  if (!sim || receiver >= sim->nodes.size() || handle < 0 || (size_t)handle >= sim->manual_notifications.size()) return -1;
  try {
    SimNode& n = sim->nodes[receiver];
    std::optional<Request> r = n.node.handle_notification(n.context, *sim->manual_notifications[handle]);
    if (should_sync) *should_sync = r ? 1 : 0;
    return 0;
  } catch (const Panic& p) { sim->error = p.msg; return -2; }
}
int lbft_oracle_node_create_request(lbft_oracle_sim* sim, uint32_t node) {
  if (!sim || node >= sim->nodes.size()) return -1;
  try {
    sim->manual_requests.push_back(sim->nodes[node].node.create_request_internal());
    return (int)sim->manual_requests.size() - 1;
  } catch (const Panic& p) { sim->error = p.msg; return -2; }
}
int lbft_oracle_node_handle_request(lbft_oracle_sim* sim, uint32_t node, int request) {
  if (!sim || node >= sim->nodes.size() || request < 0 || (size_t)request >= sim->manual_requests.size()) return -1;
  try {
    sim->manual_responses.push_back(sim->nodes[node].node.handle_request(sim->manual_requests[request]));
    return (int)sim->manual_responses.size() - 1;
  } catch (const Panic& p) { sim->error = p.msg; return -2; }
}
int lbft_oracle_node_handle_response(lbft_oracle_sim* sim, uint32_t node, int response, int64_t node_time) {
  if (!sim || node >= sim->nodes.size() || response < 0 || (size_t)response >= sim->manual_responses.size()) return -1;
  try {
    SimNode& n = sim->nodes[node];
    n.node.handle_response(n.context, sim->manual_responses[response], node_time);
    return 0;
  } catch (const Panic& p) { sim->error = p.msg; return -2; }
}
int lbft_oracle_node_view_get(const lbft_oracle_sim* sim, uint32_t node, lbft_oracle_node_view* out) {
  if (!sim || !out || node >= sim->nodes.size()) return -1;
  const SimNode& n = sim->nodes[node];
  const RecordStore& rs = *n.node.record_store;
  out->epoch_id = n.node.epoch_id;
  out->current_round = rs.current_round;
  out->highest_quorum_certificate_round = rs.highest_quorum_certificate_round;
  out->highest_timeout_certificate_round = rs.highest_timeout_certificate_round;
  out->highest_committed_round = rs.highest_committed_round;
  out->active_round = n.node.pacemaker.active_round;
  out->latest_voted_round = n.node.latest_voted_round;
  out->locked_round = n.node.locked_round;
  out->commit_count = n.context.last_committed.history.size();
  out->active_leader = n.node.pacemaker.active_leader ? (uint32_t)*n.node.pacemaker.active_leader : UINT32_MAX;
  out->election = (uint32_t)rs.election;
  out->num_current_timeouts = (uint32_t)rs.current_timeouts.size();
  out->num_current_votes = (uint32_t)rs.current_votes.size();
  out->has_proposed_block = rs.current_proposed_block ? 1 : 0;
  out->has_timeout_certificate = rs.highest_timeout_certificate ? 1 : 0;
  return 0;
}
void lbft_oracle_enable_data_writer(lbft_oracle_sim* sim) {
  sim->data_writer = true;
  sim->dw_max_round.assign(sim->nodes.size(), 0);
  sim->dw_switches.assign(sim->nodes.size(), {});
}
// round_switches.txt rows (data_writer.rs:62-86): out[round * num_nodes + node] = time or INT64_MIN (None), for
// round in [0, max_round); returns max_round.  *messages = number_of_messages.txt.
uint64_t lbft_oracle_round_switches(const lbft_oracle_sim* sim, int64_t* out, size_t cap_rounds, uint64_t* messages) {
  u64 max_round = 0;
  for (u64 r : sim->dw_max_round) max_round = std::max(max_round, r);
  size_t n = sim->nodes.size();
  for (u64 round = 0; round < max_round && round < cap_rounds; round++)
    for (size_t k = 0; k < n; k++) {
      i64 t = INT64_MIN;
      for (auto& x : sim->dw_switches[k]) if (x.first == round) { t = x.second; break; }
      out[round * n + k] = t;
    }
  if (messages) *messages = sim->dw_messages;
  return max_round;
}
uint64_t lbft_oracle_active_round(const lbft_oracle_sim* sim, uint32_t node) {
  return sim->nodes[node].node.pacemaker.active_round;
}
uint64_t lbft_oracle_epoch(const lbft_oracle_sim* sim, uint32_t node) { return sim->nodes[node].node.epoch_id; }
int64_t lbft_oracle_startup_time(const lbft_oracle_sim* sim, uint32_t node) { return sim->nodes[node].startup_time; }
void lbft_oracle_counters_get(const lbft_oracle_sim* sim, lbft_oracle_counters* out) { *out = sim->counters; }
const char* lbft_oracle_last_error(const lbft_oracle_sim* sim) { return sim->error.c_str(); }

int lbft_oracle_run_batch(const lbft_oracle_config* cfg, const uint64_t* seeds, size_t n_instances,
                          int64_t max_clock, uint32_t threads, uint32_t* commit_counts,
                          uint64_t* active_rounds, uint64_t* last_states, lbft_oracle_commit* histories,
                          size_t history_cap, lbft_oracle_counters* counters) {
  if (threads == 0) threads = 1;
  std::atomic<size_t> next{0};
  std::atomic<int> status{0};
  std::vector<lbft_oracle_counters> partial(threads);
  for (auto& p : partial) memset(&p, 0, sizeof(p));
  u32 nn = cfg->num_nodes;
  auto worker = [&](u32 tid) {
    for (;;) {
      size_t i = next.fetch_add(1);
      if (i >= n_instances) break;
      lbft_oracle_sim* sim = nullptr;
      int rc = lbft_oracle_create(cfg, seeds[i], &sim);
      if (rc == 0) rc = lbft_oracle_run_until(sim, max_clock);
      if (rc != 0) { int z = 0; status.compare_exchange_strong(z, rc); }
      if (sim) {
        for (u32 n = 0; n < nn; n++) {
          size_t o = i * nn + n;
          if (commit_counts) commit_counts[o] = (u32)lbft_oracle_commit_count(sim, n);
          if (active_rounds) active_rounds[o] = lbft_oracle_active_round(sim, n);
          if (last_states) last_states[o] = lbft_oracle_last_committed_state(sim, n);
          if (histories) lbft_oracle_committed_history(sim, n, histories + o * history_cap, history_cap);
        }
        lbft_oracle_counters& p = partial[tid];
        const lbft_oracle_counters& c = sim->counters;
        for (int k = 0; k < 4; k++) p.events[k] += c.events[k];
        p.rng_draws += c.rng_draws; p.rounds += c.rounds; p.commits += c.commits;
        p.response_inserts += c.response_inserts; p.events_scheduled += c.events_scheduled; p.saved_bytes += c.saved_bytes;
        p.max_queue = std::max(p.max_queue, c.max_queue);
        lbft_oracle_destroy(sim);
      }
    }
  };
  std::vector<std::thread> ts;
  for (u32 t = 1; t < threads; t++) ts.emplace_back(worker, t);
  worker(0);
  for (auto& t : ts) t.join();
  if (counters) {
    memset(counters, 0, sizeof(*counters));
    for (auto& p : partial) {
      for (int k = 0; k < 4; k++) counters->events[k] += p.events[k];
      counters->rng_draws += p.rng_draws; counters->rounds += p.rounds; counters->commits += p.commits;
      counters->response_inserts += p.response_inserts; counters->events_scheduled += p.events_scheduled; counters->saved_bytes += p.saved_bytes;
      counters->max_queue = std::max(counters->max_queue, p.max_queue);
    }
  }
  return status.load();
}

uint64_t lbft_oracle_siphash13(const uint8_t* bytes, size_t n) { return siphash13(bytes, n); }
void lbft_oracle_xoshiro_first(uint64_t seed, uint64_t* out, size_t n) {
  Xoshiro r(seed);
  for (size_t i = 0; i < n; i++) out[i] = r.next_u64();
}
static EpochConfiguration cfg_of(const uint64_t* w, size_t n) {
  std::vector<std::pair<Author, u64>> v;
  for (size_t i = 0; i < n; i++) v.push_back({(Author)i, w ? w[i] : 1});
  return EpochConfiguration(v);
}
uint64_t lbft_oracle_pick_author(const uint64_t* weights, size_t n, uint64_t seed) {
  return cfg_of(weights, n).pick_author(seed);
}
uint64_t lbft_oracle_leader(const uint64_t* weights, size_t n, uint64_t round) {
  Bytes b; b.u64le(round);
  return cfg_of(weights, n).pick_author(b.hash());
}
uint64_t lbft_oracle_quorum_threshold(const uint64_t* weights, size_t n) {
  return cfg_of(weights, n).quorum_threshold();
}
void lbft_oracle_sample_delays(const lbft_oracle_config* cfg, uint64_t seed, int64_t* out, size_t n) {
  RandomDelay d = RandomDelay::make(*cfg);
  Xoshiro r(seed);
  for (size_t i = 0; i < n; i++) out[i] = d.sample(r);
}
void lbft_oracle_shuffle(uint64_t seed, uint32_t* out, size_t n) {
  std::vector<u32> v(n);
  for (size_t i = 0; i < n; i++) v[i] = (u32)i;
  Xoshiro r(seed);
  r.shuffle(v);
  for (size_t i = 0; i < n; i++) out[i] = v[i];
}
double lbft_oracle_exp_strict(double x) { return lbft_exp(x, EXP_TAB); }
double lbft_oracle_log_strict(double x) { return lbft_log(x); }
// Bulk form of the arithmetic bridge between the two math modes: how many of the n points x[i] give lbft_exp(x) != the host
// libm's exp(x) (bit patterns) / an lbft_log(x) more than one ulp from the host's log (tests/test_math.py).
size_t lbft_oracle_exp_mismatches(const double* x, size_t n) {
  size_t bad = 0;
  for (size_t i = 0; i < n; i++) {
    double a = lbft_exp(x[i], EXP_TAB), b = std::exp(x[i]);
    uint64_t ua, ub;
    memcpy(&ua, &a, 8); memcpy(&ub, &b, 8);
    bad += ua != ub;
  }
  return bad;
}
size_t lbft_oracle_log_mismatches(const double* x, size_t n, size_t* off_by_one_ulp) {
  size_t bad = 0, ulp1 = 0;
  for (size_t i = 0; i < n; i++) {
    double a = lbft_log(x[i]), b = std::log(x[i]);
    int64_t ia, ib;
    memcpy(&ia, &a, 8); memcpy(&ib, &b, 8);
    int64_t d = ia > ib ? ia - ib : ib - ia;
    ulp1 += d == 1;
    bad += d > 1;
  }
  if (off_by_one_ulp) *off_by_one_ulp = ulp1;
  return bad;
}

}  // extern "C"
