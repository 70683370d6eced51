// bucket_stats.cpp -- TEST / ANALYSIS INFRASTRUCTURE (CPU only; round 6 "Stage A" of the wavefront-per-network step): how much independent work does
// ONE large network offer per calendar bucket?  A host build of the kernel logic (oracle/host_model.cpp, LBFT_HOST_BUCKETS hooks in lbft_core.h) reports
// every popped event (time, kind, node) and every scheduled one; from them, per configuration:
//   (i)   events per calendar bucket (time, kind) -- event-weighted quantiles;
//   (ii)  DISTINCT nodes per bucket, and the lanes a wavefront would fill per sub-round under two schedules:
//           (a) strict prefix: a sub-round = the longest run of consecutive events (pop order) on pairwise distinct nodes,
//           (b) per-node FIFO: sub-round j = the j-th pending event of every node of the bucket (<= 64 lanes per sub-round);
//   (iii) zero-delay hazards: events scheduled AT the current time into a bucket that pops BEFORE the rest of the current one
//         (ScheduledEvent::cmp, bft-lib/src/simulator.rs:149-161: time asc, kind desc, stamp asc), by (current kind -> new kind);
//   (iv)  what an event sends: nothing / one or two unicasts (timer aside) / a list of n - 1 (broadcast, query-all) -- the lists are the part that stays
//         RNG-ordered and serial per event (coop_bulk), the rest is what lanes can do side by side.
//   g++ -O2 -std=c++17 -Ioracle tests/tools/bucket_stats.cpp -o /tmp/bucket_stats -lpthread && /tmp/bucket_stats <config> [instances]
#define LBFT_HOST_BUCKETS 1
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#include "../../oracle/host_model.cpp"

struct Ev { int t; unsigned kind, node; unsigned pushes[4]; unsigned hazard; };
static std::vector<Ev> g_ev;
namespace lbft {
void lbft_host_pop(int t, unsigned kind, unsigned node, unsigned sender) { (void)sender; g_ev.push_back(Ev{t, kind, node, {0, 0, 0, 0}, 0}); }
void lbft_host_push(long long t, unsigned kind, unsigned node) {
  (void)node;
  if (g_ev.empty()) return;  // Simulator::new's first timers
  Ev& e = g_ev.back();
  e.pushes[kind]++;
  // pops before the rest of the current bucket: same time, 3 - kind smaller than the current event's
  if (t == (long long)e.t && kind > e.kind) e.hazard |= 1u << kind;
}
}  // namespace lbft

static double wq(std::vector<std::pair<unsigned, unsigned>>& v, double q) {  // (value, weight) -> weighted quantile
  std::sort(v.begin(), v.end());
  unsigned long long tot = 0, acc = 0;
  for (auto& x : v) tot += x.second;
  for (auto& x : v) { acc += x.second; if ((double)acc >= q * (double)tot) return x.first; }
  return v.empty() ? 0 : v.back().first;
}

int main(int argc, char** argv) {
  std::string name = argc > 1 ? argv[1] : "c5";
  size_t n_inst = argc > 2 ? (size_t)atol(argv[2]) : 2;
  lbft_oracle_config cfg;
  memset(&cfg, 0, sizeof(cfg));
  cfg.delay_model = 0; cfg.mean = 10; cfg.variance = 4; cfg.commands_per_epoch = 30000;
  cfg.target_commit_interval = 100000; cfg.delta = 20; cfg.gamma = 2.0; cfg.lambda = 0.5;
  int64_t max_clock = 300;
  std::vector<uint64_t> rights;
  auto weighted = [&]() { rights.resize(100); for (int i = 0; i < 100; i++) rights[i] = 1 + (i % 4); cfg.voting_rights = rights.data(); };
  if (name == "c4") { cfg.num_nodes = 64; cfg.variance = 400; cfg.equivocate_every = 5; }
  else if (name == "c4live") { cfg.num_nodes = 64; cfg.variance = 400; cfg.equivocate_every = 5; cfg.quirks = 3; max_clock = 1000; }
  else if (name == "c5") { cfg.num_nodes = 100; weighted(); cfg.commands_per_epoch = 50; }
  else if (name == "c5live") { cfg.num_nodes = 100; weighted(); cfg.commands_per_epoch = 3; cfg.quirks = 3; cfg.rights_rotation = 1; max_clock = 450; }
  else if (name == "c5named") { cfg.num_nodes = 100; weighted(); cfg.commands_per_epoch = 50; cfg.quirks = 3; cfg.rights_rotation = 1; max_clock = 2500; }
  else { fprintf(stderr, "config: c4 | c4live | c5 | c5live | c5named\n"); return 2; }
  const unsigned n = cfg.num_nodes;
  lbft_hostmodel_caps caps;
  memset(&caps, 0, sizeof(caps));
  caps.qcap = 8 * n * n; caps.scap = (cfg.quirks & 1) ? n * n + 8 * n : 8 * n; caps.bcap = (uint32_t)(max_clock / 10 + 64); caps.lcap = caps.bcap; caps.qheap = 1; caps.tw = 1;
  // accumulators over instances
  unsigned long long pops[4] = {0, 0, 0, 0}, total = 0, hazards[4][4] = {{0}}, sends_none = 0, sends_uni = 0, sends_list = 0, req_from_notify = 0;
  std::vector<std::pair<unsigned, unsigned>> size_w, distinct_w, lanes_a_w, lanes_b_w;  // (value, weight = events)
  unsigned long long sub_a = 0, sub_b = 0, sub_b_with_list = 0, buckets = 0;
  unsigned long long by_kind_events[4] = {0, 0, 0, 0}, by_kind_sub_b[4] = {0, 0, 0, 0};
  for (size_t inst = 0; inst < n_inst; inst++) {
    g_ev.clear();
    uint64_t seed = inst + 1;
    lbft_oracle_counters c;
    uint32_t fault = 0, mq = 0, ms = 0;
    int rc = lbft_hostmodel_run_batch(&cfg, &caps, &seed, 1, max_clock, 1, nullptr, nullptr, nullptr, nullptr, 0, &c, &fault, &mq, &ms, nullptr, nullptr, nullptr, 0);
    if (rc != 0 || fault) { fprintf(stderr, "instance %zu: rc %d fault %x\n", inst, rc, fault); return 1; }
    for (size_t i = 0; i < g_ev.size();) {
      size_t j = i;
      while (j < g_ev.size() && g_ev[j].t == g_ev[i].t && g_ev[j].kind == g_ev[i].kind) j++;
      // NOTE a bucket as popped: events appended to it while it drains (zero-delay sends into the same bucket) are part of the run [i, j)
      const unsigned S = (unsigned)(j - i), kind = g_ev[i].kind;
      std::vector<unsigned> per(n, 0);
      unsigned D = 0, M = 0;
      for (size_t k = i; k < j; k++) { if (!per[g_ev[k].node]++) D++; M = std::max(M, per[g_ev[k].node]); }
      // (a) strict prefix runs
      unsigned runs = 0;
      {
        std::vector<char> seen(n, 0);
        unsigned len = 0;
        for (size_t k = i; k < j; k++) {
          if (seen[g_ev[k].node] || len == 64) { runs++; std::fill(seen.begin(), seen.end(), 0); len = 0; }
          seen[g_ev[k].node] = 1; len++;
        }
        runs++;
      }
      // (b) per-node FIFO: sub-round r holds the nodes with > r events, 64 lanes at most
      unsigned subs = 0;
      for (unsigned r = 0; r < M; r++) { unsigned c2 = 0; for (unsigned x = 0; x < n; x++) if (per[x] > r) c2++; subs += (c2 + 63) / 64; }
      size_w.push_back({S, S}); distinct_w.push_back({D, S});
      lanes_a_w.push_back({(S + runs - 1) / runs, S}); lanes_b_w.push_back({(S + subs - 1) / subs, S});
      sub_a += runs; sub_b += subs; buckets++;
      by_kind_events[kind] += S; by_kind_sub_b[kind] += subs;
      bool list_in_bucket = false;
      for (size_t k = i; k < j; k++) {
        const Ev& e = g_ev[k];
        pops[e.kind]++; total++;
        for (unsigned q = 0; q < 4; q++) if (e.hazard & (1u << q)) hazards[e.kind][q]++;
        unsigned msgs = e.pushes[0] + e.pushes[1] + e.pushes[2];
        if (msgs == 0) sends_none++; else if (msgs <= 2) sends_uni++; else { sends_list++; list_in_bucket = true; }
        if (e.kind == 0 && e.pushes[1]) req_from_notify++;
      }
      if (list_in_bucket) sub_b_with_list += 1;
      i = j;
    }
  }
  static const char* KN[4] = {"notify", "request", "response", "timer"};
  printf("{\"config\": \"%s\", \"instances\": %zu, \"nodes\": %u, \"max_clock\": %lld, \"queue_pops_per_instance\": %.0f,\n", name.c_str(), n_inst, n, (long long)max_clock, (double)total / n_inst);
  printf(" \"pops_by_kind\": {\"notify\": %.4f, \"request\": %.4f, \"response\": %.4f, \"timer\": %.4f},\n", (double)pops[0] / total, (double)pops[1] / total, (double)pops[2] / total, (double)pops[3] / total);
  printf(" \"buckets_per_instance\": %.0f, \"events_per_bucket\": {\"mean\": %.1f, \"p10\": %.0f, \"median\": %.0f, \"p90\": %.0f},\n", (double)buckets / n_inst, (double)total / buckets,
         wq(size_w, 0.1), wq(size_w, 0.5), wq(size_w, 0.9));
  printf(" \"distinct_nodes_per_bucket\": {\"p10\": %.0f, \"median\": %.0f, \"p90\": %.0f},\n", wq(distinct_w, 0.1), wq(distinct_w, 0.5), wq(distinct_w, 0.9));
  printf(" \"lanes_per_subround\": {\"a_strict_prefix\": {\"mean\": %.1f, \"median\": %.0f}, \"b_per_node_fifo\": {\"mean\": %.1f, \"median\": %.0f}},\n", (double)total / sub_a, wq(lanes_a_w, 0.5),
         (double)total / sub_b, wq(lanes_b_w, 0.5));
  printf(" \"lanes_per_subround_b_by_kind\": {");
  for (int k = 0; k < 4; k++) printf("\"%s\": %.1f%s", KN[k], by_kind_sub_b[k] ? (double)by_kind_events[k] / by_kind_sub_b[k] : 0.0, k < 3 ? ", " : "},\n");
  printf(" \"sends\": {\"nothing_but_the_timer\": %.4f, \"one_or_two_unicasts\": %.4f, \"a_list_of_n_minus_1\": %.4f, \"buckets_with_a_list\": %.4f, \"notify_events_sending_a_request\": %.5f},\n",
         (double)sends_none / total, (double)sends_uni / total, (double)sends_list / total, (double)sub_b_with_list / buckets, (double)req_from_notify / total);
  printf(" \"zero_delay_hazards_per_event\": {");
  bool first = true;
  for (int a = 0; a < 4; a++) for (int b = 0; b < 4; b++) if (hazards[a][b]) { printf("%s\"%s->%s\": %.6f", first ? "" : ", ", KN[a], KN[b], (double)hazards[a][b] / total); first = false; }
  printf("}}\n");
  return 0;
}
